#!/usr/bin/env python3
"""GPU box: table B4 at bench.py's operating point (valid codewords, QPSK + AWGN at Es/N0 dB, the demapper's LLR map), nf frames per launch:
rate, updates per group, and the rate the never-converging launch would give for the same number of updates. With DVBS2_TIMING=1 the
library prints the cycle split (syndrome tests / sweeps) per wave.   usage: exp_awgn2.py [nf] [esn0] [reps] [table] [cap]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
esn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
table = sys.argv[4] if len(sys.argv) > 4 else "S2_TABLE_B4"; N, K, _, _ = T.ldpc_info(table)
cap = int(sys.argv[5]) if len(sys.argv) > 5 else 50
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4242)
cw = T.ldpc_encode(table, rng.integers(0, 2, (64, K), dtype=np.uint8))
n0 = 10.0 ** (-esn0 / 10.0)
tx = torch.from_numpy(np.tile((1.0 - 2.0 * cw.astype(np.float32)) * np.float32(0.5 ** 0.5), (nf // 64 + 1, 1))[:nf]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(4242)
y = tx + (n0 / 2.0) ** 0.5 * torch.randn((nf, N), generator=g, device=dev)
x = torch.clamp(torch.round(y * (2.0 * 2.0 ** 0.5 / n0)), -128, 127).to(torch.int8)
xn = torch.clamp(torch.round(torch.randn((nf, N), generator=g, device=dev) * 8.0), -128, 127).to(torch.int8)
dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=cap, outputmode=capi.OM_MESSAGE)
bits = torch.empty((nf, K // 8), dtype=torch.uint8, device=dev); ret = torch.empty((nf + 31) // 32, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(inp, n):
    for _ in range(2): dec.work_device(inp.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): dec.work_device(inp.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
ta = run(x, reps)
upd = cap - ret.cpu().numpy().astype(np.int64); upd[ret.cpu().numpy() < 0] = cap
tn = run(xn, max(2, reps // 2))
prop = nf / tn * cap / upd.mean()
print(f"awgn {table} nf={nf} Es/N0 {esn0}: {ta*1e3:.2f} ms {nf/ta:.0f} fr/s | updates/group mean {upd.mean():.2f} min {upd.min()} max {upd.max()} | noise {nf/tn:.0f} fr/s "
      f"-> proportional {prop:.0f}, frac {nf/ta/prop:.4f} | fallback rounds {dec.fallback_rounds}", flush=True)
