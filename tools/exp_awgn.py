#!/usr/bin/env python3
"""GPU box: table B4 at the operating point (valid codewords + noise, groups of 32 stop at different counts), nf frames: rate and,
with DVBS2_TIMING=1 and a timing build (DVBS2_LIB), the cycle split syndrome tests / sweeps per wave."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
table = "S2_TABLE_B4"; N, K, _, _ = T.ldpc_info(table)
base, _ = T.llr_codeword_awgn(table, 64, 4242, amp=6, sigma=5.2)
rng = np.random.default_rng(1)
llr = np.tile(base, (nf // 64 + 1, 1))[:nf]
d_in = torch.from_numpy(llr).cuda()
dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=50)
bits = torch.empty((nf, K // 8), dtype=torch.uint8, device="cuda"); ret = torch.empty(nf // 32, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): dec.work_device(d_in.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): dec.work_device(d_in.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
upd = (50 - ret.cpu().numpy()); print(f"awgn B4 nf={nf}: {dt*1e3:.2f} ms {nf/dt:.0f} fr/s, updates per group mean {upd.mean():.2f} min {upd.min()} max {upd.max()}")
