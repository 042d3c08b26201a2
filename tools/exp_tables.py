import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-dvbs2rx_amd', 'python')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from dvbs2rx_amd import LdpcDecoder, capi, ldpc_table_info
def run(table, nf=1024, trials=10, G=32):
    info = ldpc_table_info(table); N, K = info['N'], info['K']
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=trials)
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    llr = torch.clamp(torch.round(torch.randn((nf, N), generator=g, device='cuda') * 8.0), -128, 127).to(torch.int8)
    bits = torch.empty((nf, K // 8), dtype=torch.uint8, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    dec.work_device(llr.data_ptr(), nf, bits.data_ptr(), 0, 0, st)
    t_end = time.perf_counter() + float(os.environ.get('WARM_S', '0.7'))
    while time.perf_counter() < t_end:
        dec.work_device(llr.data_ptr(), nf, bits.data_ptr(), 0, 0, st)
    dt = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        dec.work_device(llr.data_ptr(), nf, bits.data_ptr(), 0, 0, st)
        torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t)
    edges = info['links_total'] * trials * nf
    print(f"{table:14s} q={info['q']:3d} conf={info['conflict_layers']:2d} nf={nf} trials={trials}: {dt*1e3:8.2f} ms  {nf/dt:9.0f} fr/s  {edges/dt/1e9:7.2f} Gedge/s  per-iter-per-frame {dt/trials/ (nf/512)*1e6:7.1f} us")
    dec.close()
for t in sys.argv[1:]:
    a = t.split(':')
    run(a[0], nf=int(a[2]) if len(a) > 2 else 1024, trials=int(a[1]) if len(a) > 1 else 10)
