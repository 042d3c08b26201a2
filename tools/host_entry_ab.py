#!/usr/bin/env python3
"""Host-buffer entry (dvbs2_ldpc_decode) rate against the resident rate, for the settings given in the environment
(DVBS2_HOST_COPY_STREAM, DVBS2_HOST_CHUNK): pageable and page-locked caller buffers, 4096- and 512-frame calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from dvbs2rx_amd import LdpcDecoder, capi, ldpc_table_info
import fec_testlib as T
dev = torch.device("cuda", 0)
info = ldpc_table_info("S2_TABLE_B4")
sizes = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,512").split(","))
host, fb = bench.host_entry(np, torch, capi, LdpcDecoder, T, dev, 0, info["N"], 4050, 4096, 50, 32, torch.cuda.current_stream().cuda_stream, 4, sizes)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("DVBS2_HOST_COPY_STREAM", "DVBS2_HOST_CHUNK") if k in os.environ) or "default"
for k, v in host.items():
    print(f"{tag:40s} {k:16s} {v['frames_per_s']:9.0f} frames/s  {v['ms_per_call']:7.2f} ms  {100 * v['frac_of_resident']:5.1f} % of resident", flush=True)
