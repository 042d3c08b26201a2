#!/usr/bin/env python3
"""GPU box: where the operating-point runs lose against the never-converging rate x cap / mean updates.
usage: opp_itemise.py <table> <esn0> <cap> <nf> <resident frames>
Bit-exact timing experiments on valid codewords + AWGN (the demapper's LLR map):
  * G = 32 (the reference's batch coupling) against G = 1 (every frame stops on its own) and the never-converging rate;
  * frames per call nf, 2 nf, 4 nf (the launch tail: the last groups run on a partly empty GPU);
  * time(cap) on never-converging input for cap = 1, 2, 4 (fixed cost per call and frame);
  * a list-scheduling model of the launch from the MEASURED update counts: groups are dispatched in order, `resident` frames fit the
    GPU; utilisation = sum(updates) / (slots x makespan): what the proportional rate cannot reach even with free dispatch."""
import os, sys, time, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
table, esn0, cap, nf0, resident = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
N, K, _, _ = T.ldpc_info(table)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(4242)
cw = T.ldpc_encode(table, rng.integers(0, 2, (64, K), dtype=np.uint8))
n0 = 10.0 ** (-esn0 / 10.0)

def awgn(nf, seed=4242):
    tx = torch.from_numpy(np.tile((1.0 - 2.0 * cw.astype(np.float32)) * np.float32(0.5 ** 0.5), (nf // 64 + 1, 1))[:nf]).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    y = tx + (n0 / 2.0) ** 0.5 * torch.randn((nf, N), generator=g, device=dev)
    return torch.clamp(torch.round(y * (2.0 * 2.0 ** 0.5 / n0)), -128, 127).to(torch.int8)

def noise(nf):
    g = torch.Generator(device=dev); g.manual_seed(1)
    return torch.clamp(torch.round(torch.randn((nf, N), generator=g, device=dev) * 8.0), -128, 127).to(torch.int8)

def run(x, G, trials, reps=5):
    nf = x.shape[0]
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=trials, outputmode=capi.OM_MESSAGE)
    bits = torch.empty((nf, K // 8), dtype=torch.uint8, device=dev); ret = torch.empty((nf + G - 1) // G, dtype=torch.int32, device=dev)
    fn = lambda: dec.work_device(x.data_ptr(), nf, bits.data_ptr(), 0, ret.data_ptr(), st)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end: fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    r = ret.cpu().numpy().astype(np.int64)
    upd = np.where(r < 0, trials, trials - r)
    fb = dec.fallback_rounds; name = dec.kernel_name
    dec.close()
    return sorted(ts)[len(ts) // 2], upd, fb, name

def model(upd, G, resident):
    """in-order list scheduling of groups onto resident / G slots; returns utilisation"""
    slots = max(1, resident // G)
    h = [0.0] * slots
    heapq.heapify(h)
    for u in upd:
        t = heapq.heappop(h); heapq.heappush(h, t + float(u))
    return float(np.sum(upd)) / (slots * max(h))

xn = noise(nf0)
tn, _, _, kname = run(xn, 32, cap, 3)
rate_n = nf0 / tn
print(f"{table} {kname}: never-converging {rate_n:.0f} fr/s at cap {cap} ({tn*1e3:.2f} ms per {nf0} frames)")
for c in (1, 2, 4):
    tc, _, _, _ = run(xn, 32, c, 5)
    print(f"  never-converging cap {c}: {tc*1e3:.3f} ms per call of {nf0} frames")
for mult in (1, 2, 4):
    nf = nf0 * mult
    if nf > 65535: break
    x = awgn(nf)
    for G in (32, 1):
        t, upd, fb, _ = run(x, G, cap)
        prop = rate_n * cap / upd.mean()
        print(f"  awgn {esn0} dB nf={nf} G={G}: {t*1e3:.2f} ms {nf/t:.0f} fr/s | updates mean {upd.mean():.2f} min {upd.min()} max {upd.max()} | "
              f"frac of proportional {nf/t/prop:.4f} | list-scheduling utilisation {model(upd, G, resident):.4f} | fallback {fb}", flush=True)
    del x
