#!/usr/bin/env python3
"""GPU box: the host-buffer entry (chunks over four streams) and two handles driven from two threads at the operating point -- does the
group-synchronous stop keep its members together when several launches share the GPU? usage: exp_awgn_host.py [frames]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
table = "S2_TABLE_B4"; N, K, _, _ = T.ldpc_info(table)
base, _ = T.llr_codeword_awgn(table, 64, 4242, amp=6, sigma=5.2)
llr = np.tile(base, (nf // 64 + 1, 1))[:nf].copy()
def rate(fn, n=3):
    fn(); t = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t) / n
dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=50)
d_in = torch.from_numpy(llr).cuda(); bits = torch.empty((nf, K // 8), dtype=torch.uint8, device="cuda"); ret = torch.empty(nf // 32, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def resident(): dec.work_device(d_in.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st); torch.cuda.synchronize()
t_res = rate(resident); print(f"resident            : {nf/t_res:9.0f} fr/s")
t_host = rate(lambda: dec.work(llr)); print(f"host entry (chunked): {nf/t_host:9.0f} fr/s  ({t_res/t_host:.3f} of resident)")
# two handles, two threads, two streams
dec2 = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=50)
s2 = torch.cuda.Stream(); d_in2 = d_in.clone(); bits2 = torch.empty_like(bits); ret2 = torch.empty_like(ret)
def both():
    def a(): dec.work_device(d_in.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st)
    def b(): dec2.work_device(d_in2.data_ptr(), nf, bits2.data_ptr(), 0, ret2.data_ptr(), s2.cuda_stream)
    ta, tb = threading.Thread(target=a), threading.Thread(target=b); ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize()
t_two = rate(both); print(f"two handles at once : {2*nf/t_two:9.0f} fr/s  ({t_res*2/t_two:.3f} of resident)")
assert torch.equal(bits, bits2) and torch.equal(ret, ret2)
