#!/usr/bin/env python3
"""GPU box: table B4 at bench.py's operating point, decoded by H handles (own state each) in a software pipeline of enqueue / finish on
H streams: does the tail of one launch (no workgroup left to dispatch while its last groups finish) disappear under the next launch, and
does the group-synchronous stop survive two sweep kernels sharing the GPU (fallback rounds)?
usage: exp_awgn_pipe.py [nf] [esn0] [reps] [handles] [prio: 0 same | 1 alternate high/low]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
esn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
H = int(sys.argv[4]) if len(sys.argv) > 4 else 2
prio = int(sys.argv[5]) if len(sys.argv) > 5 else 0
table = "S2_TABLE_B4"; N, K, _, _ = T.ldpc_info(table)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4242)
cw = T.ldpc_encode(table, rng.integers(0, 2, (64, K), dtype=np.uint8))
n0 = 10.0 ** (-esn0 / 10.0)
tx = torch.from_numpy(np.tile((1.0 - 2.0 * cw.astype(np.float32)) * np.float32(0.5 ** 0.5), (nf // 64 + 1, 1))[:nf]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(4242)
y = tx + (n0 / 2.0) ** 0.5 * torch.randn((nf, N), generator=g, device=dev)
x = torch.clamp(torch.round(y * (2.0 * 2.0 ** 0.5 / n0)), -128, 127).to(torch.int8)
xn = torch.clamp(torch.round(torch.randn((nf, N), generator=g, device=dev) * 8.0), -128, 127).to(torch.int8)
decs = [LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=50, outputmode=capi.OM_MESSAGE) for _ in range(H)]
bits = [torch.empty((nf, K // 8), dtype=torch.uint8, device=dev) for _ in range(H)]
rets = [torch.empty((nf + 31) // 32, dtype=torch.int32, device=dev) for _ in range(H)]
streams = [torch.cuda.Stream(device=dev, priority=(-1 if (prio and i % 2 == 0) else 0)) for i in range(H)]
torch.cuda.synchronize()


def run_sync(inp, n):
    d, st = decs[0], streams[0].cuda_stream
    for _ in range(2): d.work_device(inp.data_ptr(), nf, bits[0].data_ptr(), 0, rets[0].data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): d.work_device(inp.data_ptr(), nf, bits[0].data_ptr(), 0, rets[0].data_ptr(), st)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n


def run_pipe(inp, n):
    """n calls in all, round robin over the handles; call i is finished right before call i + H is enqueued."""
    def enq(i):
        h = i % H
        decs[h].enqueue_device(inp.data_ptr(), nf, bits[h].data_ptr(), 0, rets[h].data_ptr(), streams[h].cuda_stream)
    for i in range(H): enq(i)
    for i in range(H): decs[i].finish()
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n):
        if i >= H: decs[i % H].finish()
        enq(i)
    for h in range(H): decs[h].finish()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n


ts = run_sync(x, reps)
upd = 50 - rets[0].cpu().numpy().astype(np.int64); upd[rets[0].cpu().numpy() < 0] = 50
tn = run_sync(xn, max(2, reps // 2))
prop = nf / tn * 50 / upd.mean()
fb0 = sum(d.fallback_rounds for d in decs)
tp = run_pipe(x, reps * H)
ok = all(torch.equal(rets[h], rets[0]) and torch.equal(bits[h], bits[0]) for h in range(H))
fb1 = sum(d.fallback_rounds for d in decs)
tpn = run_pipe(xn, max(2, reps // 2) * H)
print(f"awgn B4 nf={nf} Es/N0 {esn0} H={H} prio={prio}: sync {nf/ts:.0f} fr/s (frac {nf/ts/prop:.4f}) | pipelined {nf/tp:.0f} fr/s (frac {nf/tp/prop:.4f}) "
      f"same results {ok} fallback rounds {fb1 - fb0} | noise sync {nf/tn:.0f} pipelined {nf/tpn:.0f} | updates mean {upd.mean():.2f}", flush=True)
