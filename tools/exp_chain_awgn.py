#!/usr/bin/env python3
"""GPU box: per-call times of the 8PSK 3/4 chain at its operating point (bench.py config3_awgn input), with and without the profiling events."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import FecChain, capi, get_fec_info, ldpc_table_info
dev = torch.device("cuda", 0); nf, G, trials = 4096, 32, 50
fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, "C3_4"); ti = ldpc_table_info(fi["table"])
n0v = np.float32(10.0 ** (-8.5 / 10.0)); rng = np.random.default_rng(31)
mb, prim = T.BCH_FIELDS[capi.FECFRAME_NORMAL]; ob = T.OracleBch(mb, prim, fi["bch_t"], fi["bch_n"])
msg0 = rng.integers(0, 256, (64, fi["bch_k"] // 8), dtype=np.uint8)
cw = T.ldpc_encode(fi["table"], np.unpackbits(ob.encode_bytes(msg0), axis=1)); rows = ti["N"] // 3
tx = T.map_8psk(np.stack([cw[:, :rows], cw[:, rows:2 * rows], cw[:, 2 * rows:]], axis=-1)).astype(np.complex64)
txd = torch.from_numpy(np.tile(tx.view(np.float32).reshape(64, -1), (nf // 64 + 1, 1))[:nf]).to(dev)
g = torch.Generator(device=dev); g.manual_seed(3131)
syms = txd + float(np.sqrt(n0v / 2.0)) * torch.randn(txd.shape, generator=g, device=dev)
ch = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=trials, device=0)
n0 = torch.tensor([float(n0v)], dtype=torch.float32, device=dev)
msg = torch.empty((nf, ch.msg_bytes), dtype=torch.uint8, device=dev); r = torch.empty(nf // G, dtype=torch.int32, device=dev); c = torch.empty(nf, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
fn = lambda: ch.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, msg.data_ptr(), r.data_ptr(), c.data_ptr(), st)
for prof in (False, True, False):
    ch.profile(prof)
    ts = []
    for _ in range(24):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print("profiling", prof, " ms per call:", " ".join(f"{x:.2f}" for x in ts))
    if prof: print("  ldpc launch ms", ch.profile(False))
# after seconds of GPU idleness (what a parity gate on the host cores leaves behind): per-call times again, and with all host cores busy meanwhile
import subprocess
for label, busy in (("idle 6 s", False), ("idle 6 s, then host cores busy during the calls", True)):
    time.sleep(6.0)
    ps = [subprocess.Popen([sys.executable, "-c", "import time\nt=time.time()\nwhile time.time()-t<4: pass"]) for _ in range(os.cpu_count() if busy else 0)]
    ts = []
    for _ in range(40):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    for p in ps: p.wait()
    print(label, " ms per call:", " ".join(f"{x:.2f}" for x in ts))
