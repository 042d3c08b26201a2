#!/usr/bin/env python3
"""bench.py -- FECFRAMEs/s of the DVB-S2 FEC decode hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; frames shard across ranks with no
   data-path collective -- "scaling": "weak", 4096 frames per GPU per step.)

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM. The headline `value`
is BASELINE config 2: QPSK 1/2 normal FECFRAMEs (DVB_S2_TABLE_B4, N=64800), LDPC capped at 50 iterations, batch 4096 per
GPU, reference batch grouping G=32, never-converging int8 LLRs clamp(round(N(0, 8^2))) so that exactly 50 updates run for
every frame (SURVEY 8(d) primary input). The same invocation also measures the other BASELINE configs and reports them
under "configs" (each with its own parity gate, timing and roofline object):
  config3  8PSK 3/4 normal: soft demapper + LDPC (S2_TABLE_B7) + BCH(48600,48408,12), 4096 frames, noise-only symbols
  config4  QPSK 1/4 short (S2_TABLE_C1), 25 iterations, 16384 frames
  config5  9/10 normal from LLRs: LDPC (S2_TABLE_B11) + BCH(58320,58192,8), 4096 frames per GPU (32768 over 8 GPUs)
  config5_s2x  S2X 154/180 normal from LLRs: LDPC (S2X_TABLE_B21) + BCH(55440,55248,12), 4096 frames per GPU
and the SURVEY 8(d) secondaries (operating points: valid codewords through the channel, groups stop at different counts; mean updates,
the roofline on the updates actually executed, the rate against the never-converging rate x cap / mean updates):
  config2_awgn  QPSK 1/2 normal, QPSK + AWGN, LLR = clamp(rint(2 sqrt(2) y / N0)) (the demapper's map), Es/N0 2.0 dB
  config3_awgn  8PSK 3/4 normal chain from SYMBOLS: 8PSK-mapped BCH o LDPC codewords + AWGN at Es/N0 8.5 dB, N0 as input (the loopback
                of examples/dvbs2_fec_ber.grc), BCH correction histogram
  config4_awgn  QPSK 1/4 short, QPSK + AWGN at Es/N0 0.5 dB (the genuine reference does not converge within 25 updates at the survey's
                -1.8 dB with the demapper's LLR scale: it fails at -0.5 dB and needs 19-22 updates at 0.0 dB)
  config2_host  the host-buffer entry dvbs2_ldpc_decode (H2D / D2H inclusive; pageable and page-locked caller buffers), and
                `pipelined`: two handles x the caller's own page-locked buffers through enqueue / finish (what a double-buffering block does)
plus `device_copy` (measured device-to-device copy bandwidth beside the 8 TB/s nominal peak), `host_link` (plain hipMemcpyAsync rates of
the box's host link: what the host entry could at most be fed with) and `roofline.mapping_ceiling` (the same kernel build on B4's
hazard-free degree-7 sibling S2X_TABLE_B3, per edge update, projected onto B4: what this mapping does when nothing orders the rows).
At N > 1 every rank additionally runs `config2_host` at the same time (per-rank and summed rates): the host feed of the sharded job.
Every parity gate compares the WHOLE batch with the genuine reference run on all host cores (--gate first: first group only).
config 1 (one frame through a CPU path) has no counterpart in the bench: the library has no CPU path by design; its
workload runs on the HIP path in tests/test_ldpc_gpu.py::test_baseline_config1_one_frame_replicated.
Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def ldpc_bytes(N, out_bytes, links_total, iters):
    # int8 LLR in + packed bits out + one int8 message read and written per edge per update
    return N + out_bytes + iters * 2 * links_total


# ------------------------------------------------------------------ CPU baseline (test infrastructure: drives the CHECKER)
def physical_cores():
    """One logical CPU per physical core among the CPUs this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, pick = set(), []
    try:
        for c in allowed:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            key = (open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip())
            if key not in seen:
                seen.add(key); pick.append(c)
    except OSError:
        pick = allowed
    return pick


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(table, N, trials, one_core_s=3.0, all_core_s=6.0):
    """The genuine reference AVX2 decoder (oracle/_ref, kind 'reference') on THIS box's host cores: one worker PROCESS per
    physical core (the reference's decoder object is a global per translation unit), all started together, a bounded wall
    time each; plus the same worker alone on one core. Falls back to the scalar restatement (kind 'port', one thread)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fec_testlib as T
    worker = os.path.join(ROOT, "tools", "cpu_ref_worker.py")
    if T.ref_ldpc() is None:
        x = T.llr_noise(32, N, 2000)
        t1 = time.perf_counter()
        T.oracle_ldpc_decode(table, x, 32, trials)
        dt = time.perf_counter() - t1
        return {"value": 32 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": f"32 frames, one scalar-port batch, noise LLRs, {trials} iterations"}

    def run(cpus, seconds):
        start = time.time() + 1.5 + 0.01 * len(cpus)  # python + numpy start-up of the workers
        procs = [subprocess.Popen([sys.executable, worker, table, str(trials), str(seconds), str(c), repr(start), str(N)],
                                  stdout=subprocess.PIPE, text=True) for c in cpus]
        frames, span = 0, 0.0
        for p in procs:
            out = p.communicate()[0].split()
            frames += int(out[0]); span = max(span, float(out[1]))
        return frames, span

    cores = physical_cores()
    f1, s1 = run(cores[:1], one_core_s)
    fa, sa = run(cores, all_core_s)
    return {"value": fa / sa, "unit": "frames/s", "cores": len(cores), "kind": "reference",
            "sample": (f"{fa} frames in {sa:.1f} s wall: {len(cores)} processes (one per physical core, pinned) each decoding AVX2 "
                       f"batches of 32 noise-LLR frames, {trials} iterations, decode only"),
            "one_core": {"value": f1 / s1, "frames": f1, "seconds": s1}, "cpu_model": cpu_model(),
            "logical_cpus": len(os.sched_getaffinity(0))}


def csrc_sha256():
    """Digest of the kernel sources (every file of gr-dvbs2rx_amd/csrc, names and contents): profiles/traffic.json records the
    tree it was profiled at (tools/gen_traffic.py) and is only believed for exactly that tree."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gr-dvbs2rx_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def measured_entry(config, kernel, frames, trials):
    """The committed PMC record (profiles/traffic.json <- tools/gen_traffic.py) of one configuration, when the profiled configuration
    (name, kernel build, batch, cap) equals the one being run AND the kernel sources are the ones that were profiled; None otherwise."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    if t.get("csrc_sha256") != csrc_sha256():
        return None
    for e in t.get("entries", []):
        if e.get("config") == config and e["kernel"] == kernel and e["frames_per_launch"] == frames and e["max_trials"] == trials:
            return e
    return None


N_SIMD, SHADER_GHZ_NOMINAL = 1024, 2.4  # 256 CUs x 4 SIMDs; the clock is MEASURED in the run (dvbs2_measure_shader_clock), this is the fallback


def valu_mix(kernel):
    """Static VALU class census of the kernel build (profiles/valu_mix.json <- tools/valu_census.py), believed only for the tree it was
    taken from: average issue cycles per VALU wave-instruction per SIMD at the measured per-class rates (full 2.65, half 4.3, quarter 8.2)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
    except Exception:
        return None
    if t.get("csrc_sha256") != csrc_sha256():
        return None
    return t.get("kernels", {}).get(kernel)


def limiter_from_counters(entry, edges_per_launch, copy_gbs=None, clock_ghz=None, kernel=None, frames_per_s=None):
    """What limits the dominant kernel, COMPUTED from the committed SQ pass of the same tree (no hard-coded text). Three fractions:
      valu_issue   SQ_INSTS_VALU x (issue cycles per instruction of the kernel's class mix, tools/valu_census.py) / all SIMD cycles of the
                   launch, with the shader clock measured in THIS run (round 5 took 4 cycles per instruction and 2.4 GHz flat);
      fabric       counter bytes / launch time against the device copy rate measured in this run;
      waiting      SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share of all wave-cycles spent waiting (barriers, LDS and memory latency, s_waitcnt).
    The verdict names the largest. `valu_issue_bound_frames_per_s`: the rate at which VALU issue alone would saturate."""
    sq = (entry or {}).get("sq")
    if not sq:
        return None
    n = max(sq["dispatches"], 1)
    ghz = clock_ghz or SHADER_GHZ_NOMINAL
    cycles = sq["dur_ns"] / n * ghz  # per launch
    insts = sq["SQ_INSTS_VALU"] / n
    mix = valu_mix(kernel) if kernel else None
    cpi = mix["cycles_per_valu_instruction"] if mix else 4.0
    issue = insts * cpi / (N_SIMD * cycles)
    out = {"source": entry.get("source"), "shader_clock_ghz": ghz, "shader_clock_measured": clock_ghz is not None,
           "issue_cycles_per_valu_instruction": cpi,
           "valu_class_mix": ({k: mix[k] / max(mix["valu"], 1) for k in ("full", "half", "quarter")} if mix else None),
           "valu_class_mix_source": "profiles/valu_mix.json (static census of the kernel text; rates tools/ubench/valu_rate.hip)" if mix else "none for this tree: 4 cycles per instruction assumed",
           "valu_issue_frac_of_simd_cycles": issue,
           "simd_cycles_per_valu_inst": N_SIMD * cycles / insts, "valu_lane_ops_per_edge": insts * 64.0 / edges_per_launch,
           "wave_cycles_waiting_frac": sq["SQ_WAIT_ANY"] / max(sq["SQ_WAVE_CYCLES"], 1),
           "wave_cycles_waiting_for_an_instruction_frac": sq["SQ_WAIT_INST_ANY"] / max(sq["SQ_WAVE_CYCLES"], 1),
           "fabric_gbs": entry["hbm_bytes_per_launch"] / (sq["dur_ns"] / n)}
    if copy_gbs:
        out["fabric_frac_of_measured_copy"] = out["fabric_gbs"] / copy_gbs
    mem = out.get("fabric_frac_of_measured_copy", out["fabric_gbs"] / HBM_PEAK_GBS)
    if frames_per_s:
        out["valu_issue_bound_frames_per_s"] = frames_per_s / max(issue, 1e-9)
    cand = {"VALU issue": issue, "memory fabric": mem, "waiting (barriers, LDS / memory latency)": out["wave_cycles_waiting_frac"]}
    top = max(cand, key=cand.get)
    out["verdict"] = (f"{top} {cand[top]:.2f} -- VALU issue {issue:.2f} of all SIMD cycles ({cpi:.2f} cycles per instruction at {ghz:.2f} GHz), fabric at {mem:.2f} of "
                      + ("the measured copy rate" if copy_gbs else "the nominal peak") + f", {out['wave_cycles_waiting_frac']:.2f} of all wave-cycles waiting")
    return out


def measured_traffic(config, kernel, frames, trials):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (measured_entry); None without a record."""
    e = measured_entry(config, kernel, frames, trials)
    return e["hbm_bytes_per_launch"] if e else None


# ------------------------------------------------------------------ timing
def timed(step, steps, warmup, shard, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    for _ in range(warmup):
        step()
    import torch
    shard.barrier_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    timed.own_s = time.perf_counter() - t0  # this rank alone, before it waits for the others (per-rank report at N > 1)
    shard.barrier_sync()
    return shard.max_over_ranks(time.perf_counter() - t0, device=dev)


def warm(fn, seconds=0.25):
    """Untimed calls for a fixed wall time: after a parity gate the GPU has been idle for seconds (the checker runs on the host cores)."""
    import torch
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()


N_REGIONS = 5


def timed_median(step, steps, shard, dev, runs, gpu_runs=None):
    """The secondary configs: N_REGIONS timed regions of K steps each (same clock discipline as the headline: barrier + synchronize on
    both sides, max over ranks), the MEDIAN region is reported and every region is listed in `ms_per_step_runs`. Beside the host clock
    every region is bracketed by two HIP events on the launch stream (`gpu_ms_per_step_runs`): a caller that lost its core shows up
    as host time without GPU time instead of being filtered out by taking a minimum (round 5 took the faster of two regions)."""
    import torch
    ts = []
    for _ in range(N_REGIONS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        holder = {}

        def region():
            if "started" not in holder:
                holder["started"] = True
                e0.record()
            step()

        t = timed(region, steps, 0, shard, dev)
        e1.record(); torch.cuda.synchronize()
        ts.append(t)
        if gpu_runs is not None:
            gpu_runs.append(e0.elapsed_time(e1) / steps)
    runs.extend(t / steps * 1e3 for t in ts)
    return sorted(ts)[len(ts) // 2]


def spread(runs):
    """(max - min) / median of the per-region times."""
    r = sorted(runs)
    return (r[-1] - r[0]) / r[len(r) // 2] if r else None


def roofline(obj, b_alg_ldpc, nf, traffic=None, config=None, trials=None, links_total=None):
    """HIP events around the dominant kernel (the LDPC sweep) on its launch stream: algorithmic bytes of one launch / its
    average duration. `limiter` from the committed SQ pass (limiter_from_counters) when one exists for this tree and config."""
    kern_ms, launches = obj.profile(False)
    avg_s = kern_ms / max(launches, 1) * 1e-3
    achieved = b_alg_ldpc * nf / avg_s / 1e9 if avg_s > 0 else 0.0
    if traffic is None and config is not None:
        traffic = measured_traffic(config, obj.kernel_name, nf, trials)
    rl = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
          "traffic": traffic, "kernel": obj.kernel_name, "avg_launch_ms": avg_s * 1e3, "launches": launches,
          "algorithmic_bytes_per_frame": b_alg_ldpc}
    if config is not None and links_total:
        rl["limiter"] = limiter_from_counters(measured_entry(config, obj.kernel_name, nf, trials), float(links_total) * trials * nf, roofline.copy_gbs,
                                              roofline.clock_ghz, obj.kernel_name, nf / avg_s if avg_s > 0 else None)
        if rl["limiter"] and rl["limiter"].get("valu_issue_bound_frames_per_s"):
            # what binds THIS algorithm on this ISA, beside the HBM fraction above: the thread-per-check-row mapping saturates the VALU
            # issue slots (and waits at its layer barriers) long before its message bytes saturate HBM
            lim = rl["limiter"]
            rl["binding"] = {"kind": "VALU issue (+ waiting at barriers), not HBM", "valu_issue_bound_frames_per_s": lim["valu_issue_bound_frames_per_s"],
                             "frac_of_valu_issue_bound": lim["valu_issue_frac_of_simd_cycles"],
                             "hbm_frac_at_the_valu_issue_bound": rl["frac"] / max(lim["valu_issue_frac_of_simd_cycles"], 1e-9)}
    return rl


roofline.copy_gbs = None  # device copy rate of this run (set in main before the first roofline object)
roofline.clock_ghz = None  # shader clock under VALU load measured in this run (dvbs2_measure_shader_clock)


def noise_llr(torch, nf, N, dev, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return torch.clamp(torch.round(torch.randn((nf, N), generator=g, device=dev) * 8.0), -128, 127).to(torch.int8)


def ldpc_gate(T, np, torch, dec, table, llr, G, trials, stream, full=True):
    """GPU output must equal the CPU checker bit for bit: the WHOLE batch against the genuine AVX2 reference on all host cores
    (full), or the first group only."""
    nf, N = (llr.shape[0] if full and T.ref_ldpc() is not None and G == 32 else G), llr.shape[1]
    d_out = torch.empty((nf, N), dtype=torch.int8, device=llr.device)
    d_b = torch.empty((nf, dec.out_bytes), dtype=torch.uint8, device=llr.device)
    d_r = torch.empty((nf + G - 1) // G, dtype=torch.int32, device=llr.device)
    dec.work_device(llr.data_ptr(), nf, d_b.data_ptr(), d_out.data_ptr(), d_r.data_ptr(), stream)
    x = llr[:nf].cpu().numpy()
    if T.ref_ldpc() is not None and G in (16, 32):
        want, wret = T.ref_ldpc_decode_parallel(table, x, 0 if G == 32 else 2, trials); who = "reference AVX2" if G == 32 else "reference generic"
    else:
        want, wret = T.oracle_ldpc_decode(table, x, G, trials); who = "oracle"
    ok = (d_r.cpu().tolist() == wret and np.array_equal(d_out.cpu().numpy(), want)
          and np.array_equal(d_b.cpu().numpy(), T.pack_bits(want, dec.message_bits)))
    if not ok:
        raise RuntimeError(f"PARITY FAILURE ({table}): GPU decode differs from the CPU checker; no number reported")
    return f"bit-exact vs {who}, {nf} of {llr.shape[0]} frames (LLRs, bits, return values)", wret


def chain_check(T, np, fi, llr_host, trials, fs, msg, ret, corr, what):
    """Chain outputs (messages, LDPC return values, BCH results) of the frames in llr_host vs the CPU chain."""
    want_msg, want_corr, wret, who = T.chain_expect(fi["table"], fi["bch_n"], fi["bch_t"], fs, llr_host, trials)
    nfr = llr_host.shape[0]
    ok = (ret[:len(wret)].cpu().tolist() == wret and corr[:nfr].cpu().numpy().tolist() == want_corr.tolist()
          and np.array_equal(msg[:nfr].cpu().numpy(), want_msg))
    if not ok:
        raise RuntimeError(f"PARITY FAILURE (chain {fi['table']}): GPU chain differs from the CPU checker")
    return f"bit-exact vs {what}{who}, {nfr} frames (messages, BCH results, LDPC return values)"


def device_copy_bandwidth(torch, dev):
    """Measured device-to-device copy rate (read + write bytes per second): the practical HBM ceiling beside the nominal peak."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    a.fill_(1)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return {"bytes_copied": n, "ms": ms, "read_plus_write_gbs": 2 * n / ms / 1e6, "nominal_peak_gbs": HBM_PEAK_GBS,
            "frac_of_nominal": 2 * n / ms / 1e6 / HBM_PEAK_GBS}


def host_link_bandwidth(capi, local):
    """Plain hipMemcpyAsync rate of this box's host link (dvbs2_measure_host_copy: no decode, no pipeline): 1 GiB of hipHostMalloc
    memory over one and four streams, hipHostRegister'ed malloc memory (what dvbs2_host_register gives a caller), pageable memory."""
    import ctypes as C
    out = {"bytes": 1 << 30, "unit": "GB/s"}
    for name, nbytes, streams, kind in (("hipHostMalloc_1_stream", 1 << 30, 1, 0), ("hipHostMalloc_4_streams", 1 << 30, 4, 0),
                                        ("hipHostRegister_1_stream", 1 << 30, 1, 1), ("pageable_1_stream", 1 << 28, 1, 2)):
        h2d, d2h = C.c_double(), C.c_double()
        capi.check(capi.lib.dvbs2_measure_host_copy(local, nbytes, streams, kind, C.byref(h2d), C.byref(d2h)))
        out[name] = {"h2d": h2d.value, "d2h": d2h.value}
    return out


def host_entry(np, torch, capi, LdpcDecoder, T, dev, local, N, out_bytes, nf, trials, G, stream, steps2, sizes, shard=None):
    """dvbs2_ldpc_decode (host buffers in and out) beside the resident rate of the same frames: pageable and page-locked caller buffers.
    With `shard` (N > 1) every rank starts its calls together, so that the ranks' transfers compete for the host like in a receiver."""
    d = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", outputmode=capi.OM_MESSAGE,
                    max_trials=trials, group_size=G, max_frames=nf, device=local)
    host = {}
    for frames in sizes:
        xp = noise_llr(torch, frames, N, dev, 12345).cpu().numpy()
        for mode in ("pageable", "page_locked"):
            # page_locked: the caller's buffers are driver-allocated page-locked memory (dvbs2_host_alloc = hipHostMalloc), what a GNU Radio custom
            # buffer would be. (Until round 6 this leg registered numpy's heap memory with dvbs2_host_register: on the round's boxes -- Linux 6.18,
            # transparent_hugepage=always -- the D2H copy into such a registration faulted intermittently, "write access to a read-only page".)
            hb = []
            if mode == "page_locked":
                from dvbs2rx_amd import HostBuffer
                hb = [HostBuffer(xp.shape, np.int8), HostBuffer((frames, out_bytes), np.uint8), HostBuffer(((frames + G - 1) // G,), np.int32)]
                xh, bits_h, ret_h = (h.array for h in hb)
                xh[...] = xp
            else:
                xh = xp; bits_h = np.empty((frames, out_bytes), np.uint8); ret_h = np.empty((frames + G - 1) // G, np.int32)
            call = lambda: capi.check(capi.lib.dvbs2_ldpc_decode(d._h, xh.ctypes.data, frames, trials, capi.OM_MESSAGE,
                                                                 bits_h.ctypes.data, None, ret_h.ctypes.data))
            if os.environ.get("BENCH_TRACE"):
                print(f"[bench]   host_entry {frames} {mode} x {xh.ctypes.data:#x} bits {bits_h.ctypes.data:#x}+{bits_h.nbytes:#x} ret {ret_h.ctypes.data:#x}", file=sys.stderr, flush=True)
            call()
            if shard is not None:
                shard.barrier_sync()
            t0 = time.perf_counter()
            for _ in range(steps2):
                call()
            th = (time.perf_counter() - t0) / steps2
            if shard is not None:
                shard.barrier_sync()
            if os.environ.get("BENCH_TRACE"):
                print(f"[bench]   host_entry {frames} {mode}: host calls done, resident run", file=sys.stderr, flush=True)
            dx = torch.from_numpy(xh).to(dev); db = torch.empty((frames, out_bytes), dtype=torch.uint8, device=dev)
            fnr = lambda: d.work_device(dx.data_ptr(), frames, db.data_ptr(), 0, 0, stream)
            fnr(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps2):
                fnr()
            torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / steps2
            if not np.array_equal(bits_h, db.cpu().numpy()):
                raise RuntimeError("PARITY FAILURE: host-buffer entry differs from the device entry")
            host[f"{frames}_{mode}"] = {"frames_per_call": frames, "frames_per_s": frames / th, "ms_per_call": th * 1e3,
                                        "resident_frames_per_s": frames / tr, "frac_of_resident": tr / th,
                                        # bytes the call moves over the link per second of the call (in + out): the rate the decode
                                        # CONSUMES, not the link's capacity -- that is `host_link`
                                        "link_gbs_consumed": frames * (N + out_bytes + 4.0 / G) / th / 1e9}
            del dx, db, xh, bits_h, ret_h
            for h in hb:
                h.free()
    fb = d.fallback_rounds
    d.close()
    return host, fb


def host_pipelined(np, torch, capi, LdpcDecoder, dev, local, N, out_bytes, nf, trials, G, sizes, sync_calls):
    """The host feed as a double-buffering block runs it: TWO handles, each with its own stream and the caller's own
    page-locked input / output buffers; per call H2D copy -> dvbs2_ldpc_enqueue_device -> D2H copies on the handle's stream, call i is
    finished (dvbs2_ldpc_finish) right before call i + 2 is issued, so the copies and the launch tail of one call run under the other
    call's decode. Results are compared with the synchronous host entry's resident run of the same frames."""
    out = {"what": "two handles x (own stream, page-locked host buffers of the caller): H2D + enqueue + D2H per call, call i finished right "
                   "before call i + 2 is issued; beside the synchronous dvbs2_ldpc_decode rates in `calls`"}
    for frames in sizes:
        hs = [LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", outputmode=capi.OM_MESSAGE,
                          max_trials=trials, group_size=G, max_frames=frames, device=local) for _ in range(2)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(2)]
        x0 = noise_llr(torch, frames, N, dev, 12345)
        hx = [x0.cpu().pin_memory() for _ in range(2)]
        dx = [torch.empty_like(x0) for _ in range(2)]
        db = [torch.empty((frames, out_bytes), dtype=torch.uint8, device=dev) for _ in range(2)]
        dr = [torch.empty((frames + G - 1) // G, dtype=torch.int32, device=dev) for _ in range(2)]
        hb = [torch.empty((frames, out_bytes), dtype=torch.uint8).pin_memory() for _ in range(2)]
        hr = [torch.empty((frames + G - 1) // G, dtype=torch.int32).pin_memory() for _ in range(2)]

        def issue(i):
            k = i % 2
            with torch.cuda.stream(sts[k]):
                dx[k].copy_(hx[k], non_blocking=True)
                hs[k].enqueue_device(dx[k].data_ptr(), frames, db[k].data_ptr(), 0, dr[k].data_ptr(), sts[k].cuda_stream)
                hb[k].copy_(db[k], non_blocking=True); hr[k].copy_(dr[k], non_blocking=True)

        def done(i):
            hs[i % 2].finish(); sts[i % 2].synchronize()

        ncalls = 12 if frames >= 2048 else 48

        def pipe():
            for i in range(ncalls):
                if i >= 2:
                    done(i)
                issue(i)
            done(0); done(1)

        issue(0); issue(1); done(0); done(1)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); pipe(); tp = time.perf_counter() - t0
        # the same frames, resident, synchronous (one handle)
        fnr = lambda: hs[0].work_device(x0.data_ptr(), frames, db[0].data_ptr(), 0, dr[0].data_ptr(), sts[0].cuda_stream)
        fnr(); torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(3):
            fnr()
        torch.cuda.synchronize(dev); tr = (time.perf_counter() - t0) / 3
        same = bool(torch.equal(hb[0], hb[1]) and torch.equal(hb[0], db[0].cpu()) and torch.equal(hr[0], dr[0].cpu()))
        if not same:
            raise RuntimeError("PARITY FAILURE: pipelined host feed differs from the device entry")
        sync = sync_calls.get(f"{frames}_page_locked", {})
        out[str(frames)] = {"frames_per_call": frames, "calls": ncalls, "frames_per_s": frames * ncalls / tp, "resident_frames_per_s": frames / tr,
                            "frac_of_resident": (frames * ncalls / tp) / (frames / tr), "same_results": same,
                            "synchronous_page_locked_frac_of_resident": sync.get("frac_of_resident"),
                            "fallback_rounds": hs[0].fallback_rounds + hs[1].fallback_rounds}
        for h in hs:
            h.close()
    return out


def pipelined_two(torch, handles, enq, shard, dev, ncalls=12):
    """Two handles in a software pipeline (enqueue / finish, one stream each, call i finished right before call i + 2 is enqueued): what a
    double-buffering block does. enq(k, stream) enqueues one call on handle k. Returns (seconds for ncalls calls, ncalls)."""
    sts = [torch.cuda.Stream(device=dev) for _ in handles]

    def pipe():
        for i in range(ncalls):
            if i >= 2:
                handles[i % 2].finish()
            enq(i % 2, sts[i % 2].cuda_stream)
        handles[0].finish(); handles[1].finish()

    enq(0, sts[0].cuda_stream); enq(1, sts[1].cuda_stream); handles[0].finish(); handles[1].finish()
    torch.cuda.synchronize(dev)
    ts = sorted(timed(pipe, 1, 0, shard, dev) for _ in range(3))
    return ts[1], ncalls, [t / ncalls * 1e3 for t in ts]


def demap_entry(np, torch, capi, Demapper, T, dev, local, nf, stream, copy_gbs):
    """The standalone soft demapper (what the UNFUSED block path runs: lib/xfecframe_demapper_cb_impl.cc:152-176 / lib/qpsk.h:208-214):
    HIP events around 10 launches of 4096 normal frames; algorithmic bytes per frame = 8 bytes per symbol in + one int8 LLR per bit out
    = 8 N / n_mod + N. (In the chains the demapper is fused into the sweep kernel's frame load and does not exist as a kernel.)"""
    out = {}
    for name, rate, mod, order_m in (("qpsk_1_2_normal", "C1_2", capi.MOD_QPSK, 4), ("8psk_3_4_normal", "C3_4", capi.MOD_8PSK, 8)):
        dm = Demapper(framesize=capi.FECFRAME_NORMAL, rate=rate, constellation=mod, max_frames=nf, device=local)
        g = torch.Generator(device=dev); g.manual_seed(99)
        syms = torch.randn((nf, dm.n_syms * 2), generator=g, device=dev) * 0.7071
        n0 = torch.tensor([0.2], dtype=torch.float32, device=dev)
        llr = torch.empty((nf, dm.n_llr), dtype=torch.int8, device=dev)
        fn = lambda: dm.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, llr.data_ptr(), stream)
        fn(); torch.cuda.synchronize()
        want = T.oracle_demap(syms[:4].cpu().numpy().view(np.complex64), np.float32(0.2), order_m, dm.column_order)
        if not np.array_equal(llr[:4].cpu().numpy(), want):
            raise RuntimeError("PARITY FAILURE: demapper differs from its restatement")
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        b = 8 * dm.n_syms + dm.n_llr
        gbs = b * nf / ms / 1e6
        out[name] = {"frames": nf, "ms_per_launch": ms, "frames_per_s": nf / ms * 1e3, "algorithmic_bytes_per_frame": b, "achieved_gbs": gbs,
                     "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "frac_of_measured_copy": (gbs / copy_gbs) if copy_gbs else None,
                     "parity": "bit-exact vs the demapper restatement (parity unpinned: VOLK absent), first 4 frames"}
        dm.close(); del syms, llr
    return out


def chain_host_entry(np, torch, capi, FecChain, dev, local, nf, trials, G, stream, syms_dev, n0v, sizes, link_h2d_gbs, steps2, what):
    """dvbs2_chain_decode (HOST symbols in, HOST message bytes out: H2D + demapper + LDPC + BCH + D2H per synchronous call) beside the
    device-resident chain on the same frames: pageable and page-locked caller buffers; and `pipelined`: two handles x the caller's own
    page-locked buffers through H2D copy + dvbs2_chain_enqueue_device + D2H copy / dvbs2_chain_finish. An 8PSK normal frame is
    172 800 B of symbols: the host link bounds this entry (`link_bound_frames_per_s`)."""
    ch = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=trials, device=local)
    ns, mb = ch.n_syms, ch.msg_bytes
    bytes_per_frame = ns * 8 + mb + 4 + 4.0 / G
    res = {"what": what, "bytes_over_the_link_per_frame": bytes_per_frame,
           "link_bound_frames_per_s": (link_h2d_gbs * 1e9 / (ns * 8)) if link_h2d_gbs else None, "calls": {}}
    n0h = np.array([n0v], np.float32)
    d_n0 = torch.tensor([float(n0v)], dtype=torch.float32, device=dev)
    for frames in sizes:
        sh = syms_dev[:frames].cpu().numpy()
        d_msg = torch.empty((frames, mb), dtype=torch.uint8, device=dev)
        d_ret = torch.empty((frames + G - 1) // G, dtype=torch.int32, device=dev)
        d_corr = torch.empty(frames, dtype=torch.int32, device=dev)
        fnr = lambda: ch.work_device(syms_dev.data_ptr(), frames, d_n0.data_ptr(), 1, d_msg.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(), stream)
        fnr(); torch.cuda.synchronize()
        trs = []
        for _ in range(max(3, steps2)):
            t0 = time.perf_counter(); fnr(); torch.cuda.synchronize(); trs.append(time.perf_counter() - t0)
        tr = sorted(trs)[len(trs) // 2]
        want_msg, want_corr = d_msg.cpu().numpy(), d_corr.cpu().numpy()
        sp = sh
        for mode in ("pageable", "page_locked"):
            hb = []
            if mode == "page_locked":  # driver-allocated page-locked buffers of the caller (dvbs2_host_alloc; see host_entry)
                from dvbs2rx_amd import HostBuffer
                hb = [HostBuffer(sp.shape, sp.dtype), HostBuffer((frames, mb), np.uint8), HostBuffer(((frames + G - 1) // G,), np.int32), HostBuffer((frames,), np.int32)]
                sh, msg_h, ret_h, corr_h = (h.array for h in hb)
                sh[...] = sp
            else:
                sh = sp; msg_h = np.empty((frames, mb), np.uint8); ret_h = np.empty((frames + G - 1) // G, np.int32); corr_h = np.empty(frames, np.int32)
            call = lambda: ch.work_host_ptr(sh.ctypes.data, frames, n0h.ctypes.data, 1, msg_h.ctypes.data, ret_h.ctypes.data, corr_h.ctypes.data)
            call()
            ths = []
            for _ in range(max(3, steps2)):
                t0 = time.perf_counter(); call(); ths.append(time.perf_counter() - t0)
            th = sorted(ths)[len(ths) // 2]
            if not (np.array_equal(msg_h, want_msg) and np.array_equal(corr_h, want_corr)):
                raise RuntimeError("PARITY FAILURE: host-pointer chain entry differs from the device entry")
            res["calls"][f"{frames}_{mode}"] = {
                "frames_per_call": frames, "frames_per_s": frames / th, "ms_per_call": th * 1e3, "ms_per_call_runs": [t * 1e3 for t in ths],
                "resident_frames_per_s": frames / tr, "frac_of_resident": tr / th, "link_gbs_consumed": frames * bytes_per_frame / th / 1e9,
                "frac_of_link_bound": (frames / th) / res["link_bound_frames_per_s"] if res["link_bound_frames_per_s"] else None}
            del sh, msg_h, ret_h, corr_h
            for h in hb:
                h.free()
        del d_msg, d_ret, d_corr
    ch.close()
    # two handles x page-locked buffers of the caller, device-pointer ABI
    pipe = {}
    for frames in sizes:
        hs = [FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=frames, max_trials=trials, device=local) for _ in range(2)]
        hx = [syms_dev[:frames].cpu().pin_memory() for _ in range(2)]
        dx = [torch.empty((frames, ns * 2), dtype=torch.float32, device=dev) for _ in range(2)]
        dm = [torch.empty((frames, mb), dtype=torch.uint8, device=dev) for _ in range(2)]
        dc = [torch.empty(frames, dtype=torch.int32, device=dev) for _ in range(2)]
        hm = [torch.empty((frames, mb), dtype=torch.uint8).pin_memory() for _ in range(2)]
        hc = [torch.empty(frames, dtype=torch.int32).pin_memory() for _ in range(2)]
        sts = [torch.cuda.Stream(device=dev) for _ in range(2)]

        def issue(i):
            k = i % 2
            with torch.cuda.stream(sts[k]):
                dx[k].copy_(hx[k], non_blocking=True)
                hs[k].enqueue_device(dx[k].data_ptr(), frames, d_n0.data_ptr(), 1, dm[k].data_ptr(), 0, dc[k].data_ptr(), sts[k].cuda_stream)
                hm[k].copy_(dm[k], non_blocking=True); hc[k].copy_(dc[k], non_blocking=True)

        def done(i):
            hs[i % 2].finish(); sts[i % 2].synchronize()

        ncalls = 8 if frames >= 2048 else 32

        def run_pipe():
            for i in range(ncalls):
                if i >= 2:
                    done(i)
                issue(i)
            done(0); done(1)

        issue(0); issue(1); done(0); done(1)
        torch.cuda.synchronize(dev)
        tps = []
        for _ in range(3):
            t0 = time.perf_counter(); run_pipe(); tps.append(time.perf_counter() - t0)
        tp = sorted(tps)[1]
        sync = res["calls"].get(f"{frames}_page_locked", {})
        same = bool(torch.equal(hm[0], hm[1]))
        pipe[str(frames)] = {"frames_per_call": frames, "calls": ncalls, "frames_per_s": frames * ncalls / tp,
                             "frac_of_resident": (frames * ncalls / tp) / sync.get("resident_frames_per_s", float("nan")),
                             "frac_of_link_bound": (frames * ncalls / tp) / res["link_bound_frames_per_s"] if res["link_bound_frames_per_s"] else None,
                             "link_gbs_consumed": frames * ncalls * bytes_per_frame / tp / 1e9, "same_results": same}
        for h in hs:
            h.close()
        del hx, dx, dm, dc, hm, hc
    pipe["what"] = ("two handles x the caller's own page-locked buffers, WHOLE-call copies: H2D + dvbs2_chain_enqueue_device + D2H per call, call i finished right before call i + 2 "
                    "is issued; the two calls' input copies share the host link, so this form is NOT faster than the synchronous entry, which overlaps copies and kernels chunk by chunk itself")
    res["pipelined"] = pipe
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--trials", type=int, default=50)
    ap.add_argument("--group", type=int, default=32)
    ap.add_argument("--input", choices=["noise", "awgn"], default="noise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--only", default="", help="comma list of extra configs to run (default: all at N=1, config5 at N>1)")
    ap.add_argument("--gate", choices=["full", "first", "none"], default="full",
                    help="parity gates: whole batch vs the genuine reference on all host cores / first group / none (profiling passes)")
    ap.add_argument("--esn0", type=float, default=2.0, help="Es/N0 in dB of the operating-point run (config2_awgn)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from dvbs2rx_amd import Demapper, FecChain, LdpcDecoder, capi, get_fec_info, ldpc_table_info, shard

    world, rank, local = shard.init_from_env()  # nccl (= RCCL) rendezvous when WORLD_SIZE > 1

    def trace(stage):  # BENCH_TRACE=1: stage names on stderr as they start (a GPU fault aborts the process without a Python traceback)
        if os.environ.get("BENCH_TRACE"):
            print(f"[bench] {stage}", file=sys.stderr, flush=True)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if capi.lib.dvbs2_device_count() < 1:
        raise RuntimeError("no HIP device: the hot path has no CPU fallback")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fec_testlib as T  # the CPU checkers: parity gates and the cpu_baseline leg only
    stream = torch.cuda.current_stream().cuda_stream
    G = args.group

    # ---------------------------------------------------------------- headline: BASELINE config 2
    table = "S2_TABLE_B4"
    info = ldpc_table_info(table)
    N, K = info["N"], info["K"]
    nf = args.frames
    gate_on, gate_full = args.gate != "none", args.gate == "full"
    dec = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2",
                      outputmode=capi.OM_MESSAGE, max_trials=args.trials, group_size=G, max_frames=nf, device=local)
    out_bytes = dec.out_bytes

    def qpsk_awgn_llr(tbl, frames, esn0_db, seed):
        """SURVEY 8(d) secondary input: valid codewords (64 distinct, CPU encoder of the test library), QPSK symbols
        (1 - 2c) / sqrt(2) per dimension + AWGN of variance N0 / 2, fresh noise for every frame, through the demapper's map
        LLR = clamp(rint(2 sqrt(2) y / N0)) (lib/qpsk.h:208-214)."""
        ti = ldpc_table_info(tbl)
        rng = np.random.default_rng(seed)
        cw = T.ldpc_encode(tbl, rng.integers(0, 2, (64, ti["K"]), dtype=np.uint8))
        n0 = 10.0 ** (-esn0_db / 10.0)
        tx = torch.from_numpy(np.tile((1.0 - 2.0 * cw.astype(np.float32)) * np.float32(0.5 ** 0.5), (frames // 64 + 1, 1))[:frames]).to(dev)
        g = torch.Generator(device=dev); g.manual_seed(seed)
        y = tx + (n0 / 2.0) ** 0.5 * torch.randn((frames, ti["N"]), generator=g, device=dev)
        return torch.clamp(torch.round(y * (2.0 * 2.0 ** 0.5 / n0)), -128, 127).to(torch.int8)

    def awgn_llr(seed):
        return qpsk_awgn_llr(table, nf, args.esn0, seed)

    def awgn_8psk_symbols(es_db):
        """config 3's operating-point input: valid BCH o LDPC codewords of 8PSK 3/4 normal (64 distinct), 8PSK-mapped through the inverse of
        the block's de-interleaver, + AWGN at Es/N0 = es_db, fresh noise per frame. Returns (symbols on the device, N0, the 64 messages)."""
        fi3 = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, "C3_4")
        ti3 = ldpc_table_info(fi3["table"])
        n0v_ = np.float32(10.0 ** (-es_db / 10.0))
        rng = np.random.default_rng(31)
        mb, prim = T.BCH_FIELDS[capi.FECFRAME_NORMAL]
        ob = T.OracleBch(mb, prim, fi3["bch_t"], fi3["bch_n"])
        msg0_ = rng.integers(0, 256, (64, fi3["bch_k"] // 8), dtype=np.uint8)
        cw = T.ldpc_encode(fi3["table"], np.unpackbits(ob.encode_bytes(msg0_), axis=1))
        rows = ti3["N"] // 3
        tx = T.map_8psk(np.stack([cw[:, :rows], cw[:, rows:2 * rows], cw[:, 2 * rows:]], axis=-1)).astype(np.complex64)
        txd = torch.from_numpy(np.tile(tx.view(np.float32).reshape(64, -1), (nf // 64 + 1, 1))[:nf]).to(dev)
        g = torch.Generator(device=dev); g.manual_seed(3131)
        sy = txd + float(np.sqrt(n0v_ / 2.0)) * torch.randn(txd.shape, generator=g, device=dev)
        return sy, n0v_, msg0_

    trace("device_copy")
    device_copy = device_copy_bandwidth(torch, dev)  # first: the limiter objects compare the fabric-side rate with it
    roofline.copy_gbs = device_copy["read_plus_write_gbs"]
    shader_clock = None
    try:
        import ctypes as C
        ghz, pms = C.c_double(), C.c_double()
        capi.check(capi.lib.dvbs2_measure_shader_clock(local, C.byref(ghz), C.byref(pms)))
        shader_clock = {"ghz": ghz.value, "probe_kernel_ms": pms.value,
                        "what": "s_memtime delta / s_memrealtime delta (100 MHz) of a ~1 ms kernel that keeps every SIMD issuing VALU adds"}
        roofline.clock_ghz = ghz.value
    except Exception as e:  # (diagnostic only)
        shader_clock = {"error": str(e)}
    trace("headline input + gate")
    llr = noise_llr(torch, nf, N, dev, 12345 + rank) if args.input == "noise" else awgn_llr(4242 + rank)
    d_bits = torch.empty((nf, out_bytes), dtype=torch.uint8, device=dev)
    d_ret = torch.empty((nf + G - 1) // G, dtype=torch.int32, device=dev)
    parity = ldpc_gate(T, np, torch, dec, table, llr, G, args.trials, stream, gate_full)[0] if rank == 0 and gate_on else "skipped"

    def step():
        dec.work_device(llr.data_ptr(), nf, d_bits.data_ptr(), 0, d_ret.data_ptr(), stream)

    trace("headline timed region")
    for _ in range(args.warmup):
        step()
    dec.profile(True)  # HIP events around the dominant kernel, on its launch stream
    dt = timed(step, args.steps, 0, shard, dev)
    own_s = timed.own_s
    iters_mean = float(torch.where(d_ret < 0, torch.full_like(d_ret, args.trials), args.trials - d_ret).float().mean().item())
    b_alg = ldpc_bytes(N, out_bytes, info["links_total"], iters_mean)
    rl = roofline(dec, b_alg, nf, None, "config2" if args.input == "noise" else None, args.trials, info["links_total"])
    fps = world * nf * args.steps / dt
    out = {
        "metric": "FECFRAMEs/sec (coded Gbit/s) @ 50 LDPC iters, QPSK 1/2 normal",
        "value": fps, "unit": "frames/s", "coded_gbps": fps * N / 1e9,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8", "data": "synthetic",
        "config": {"workload": f"QPSK 1/2 normal FECFRAME (DVB_S2_TABLE_B4, N=64800), {args.trials} LDPC iterations cap, "
                               f"batch={nf} frames per GPU, group G={G}, input={args.input}",
                   "frames_per_gpu": nf, "max_trials": args.trials, "group_size": G,
                   "mean_iterations": iters_mean, "parallelism": f"frames sharded over {world} GPU(s), no collective"},
        "parity": parity, "roofline": rl,
    }
    if world > 1:
        # every rank's own rate over the same K steps (its clock from the common start to ITS last step's completion): the spread
        # shows whether the ranks run evenly or one is held back (host feed, a slower device)
        per_rank = shard.gather_over_ranks(nf * args.steps / own_s, device=dev)
        out["per_rank"] = {"frames_per_s": per_rank, "min": min(per_rank), "max": max(per_rank),
                           "spread": (max(per_rank) - min(per_rank)) / max(per_rank),
                           "note": "each rank's own clock over the same K steps (common start after the barrier, its own last "
                                   "completion); `value` uses the slowest rank's time"}
    out["shader_clock"] = shader_clock
    out["fallback_rounds"] = dec.fallback_rounds  # host-driven rounds of the group stop (zero in normal operation)
    dec.close()
    del llr, d_bits

    # ---------------------------------------------------------------- the other BASELINE configs, same clock discipline
    steps2 = max(2, min(args.steps, 3))
    want = [c for c in args.only.split(",") if c] or (["config3", "config4", "config5", "config5_s2x"] if world == 1 else ["config5"])
    configs = {}

    def ldpc_only(name, tbl, frames, trials, label):
        trace(name)
        ti = ldpc_table_info(tbl)
        d = LdpcDecoder(table=tbl, message_bits=ti["K"], outputmode=capi.OM_MESSAGE, max_trials=trials, group_size=G,
                        max_frames=frames, device=local)
        x = noise_llr(torch, frames, ti["N"], dev, 777 + rank)
        b = torch.empty((frames, d.out_bytes), dtype=torch.uint8, device=dev)
        r = torch.empty((frames + G - 1) // G, dtype=torch.int32, device=dev)
        par = ldpc_gate(T, np, torch, d, tbl, x, G, trials, stream, gate_full)[0] if rank == 0 and gate_on else "skipped"
        fn = lambda: d.work_device(x.data_ptr(), frames, b.data_ptr(), 0, r.data_ptr(), stream)
        warm(fn); d.profile(True)
        runs, gruns = [], []; t = timed_median(fn, steps2, shard, dev, runs, gruns)
        bl = ldpc_bytes(ti["N"], d.out_bytes, ti["links_total"], trials)
        configs[name] = {"workload": label, "value": world * frames * steps2 / t, "unit": "frames/s",
                         "coded_gbps": world * frames * steps2 / t * ti["N"] / 1e9, "frames_per_gpu": frames, "max_trials": trials,
                         "steps": steps2, "ms_per_step": t / steps2 * 1e3, "ms_per_step_runs": runs, "gpu_ms_per_step_runs": gruns, "ms_per_step_spread": spread(runs), "parity": par, "roofline": roofline(d, bl, frames, None, name, trials, ti["links_total"])}
        d.close()

    def llr_chain(name, rate, frames, trials, label):
        trace(name)
        fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, rate)
        ti = ldpc_table_info(fi["table"])
        ch = FecChain(rate=rate, group_size=G, max_frames=frames, max_trials=trials, device=local, from_llr=True)
        x = noise_llr(torch, frames, ti["N"], dev, 888 + rank)
        m = torch.empty((frames, ch.msg_bytes), dtype=torch.uint8, device=dev)
        r = torch.empty((frames + G - 1) // G, dtype=torch.int32, device=dev)
        c = torch.empty(frames, dtype=torch.int32, device=dev)
        fn = lambda: ch.work_llr_device(x.data_ptr(), frames, m.data_ptr(), r.data_ptr(), c.data_ptr(), stream)
        fn()
        par = "skipped"
        if rank == 0 and gate_on:
            ng = frames if gate_full and T.ref_ldpc() is not None and G == 32 else G
            par = chain_check(T, np, fi, x[:ng].cpu().numpy(), trials, capi.FECFRAME_NORMAL, m, r, c, "")
        warm(fn)  # (the checker kept the GPU idle for seconds)
        ch.profile(True)
        runs, gruns = [], []; t = timed_median(fn, steps2, shard, dev, runs, gruns)
        bl = ldpc_bytes(ti["N"], fi["bch_n"] // 8, ti["links_total"], trials)
        b_step = bl + fi["bch_n"] // 8 + fi["bch_k"] // 8
        val = world * frames * steps2 / t
        configs[name] = {"workload": label, "value": val, "unit": "frames/s", "coded_gbps": val * ti["N"] / 1e9,
                         "frames_per_gpu": frames, "frames_total": world * frames, "max_trials": trials, "steps": steps2,
                         "ms_per_step": t / steps2 * 1e3, "ms_per_step_runs": runs, "gpu_ms_per_step_runs": gruns, "ms_per_step_spread": spread(runs), "parity": par, "roofline": roofline(ch, bl, frames, None, name, trials, ti["links_total"]),
                         "step_bytes_per_frame": b_step, "step_frac_of_hbm_peak": b_step * val / world / 1e9 / HBM_PEAK_GBS}
        ch.close()

    if not args.no_configs:
        if "config3" in want:
            trace("config3")
            fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, "C3_4")
            ch = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=args.trials, device=local)
            g = torch.Generator(device=dev); g.manual_seed(777 + rank)
            syms = torch.randn((nf, ch.n_syms * 2), generator=g, device=dev) * 0.7071
            n0 = torch.tensor([1.0], dtype=torch.float32, device=dev)
            msg = torch.empty((nf, ch.msg_bytes), dtype=torch.uint8, device=dev)
            r = torch.empty((nf + G - 1) // G, dtype=torch.int32, device=dev)
            c = torch.empty(nf, dtype=torch.int32, device=dev)
            fn = lambda: ch.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, msg.data_ptr(), r.data_ptr(), c.data_ptr(), stream)
            fn()
            par = "skipped"
            if rank == 0 and gate_on:  # the CPU chain: demapper restatement -> genuine LDPC on all cores -> BCH codec
                ng = nf if gate_full and T.ref_ldpc() is not None and G == 32 else G
                x = T.oracle_demap(syms[:ng].cpu().numpy().view(np.complex64), np.float32(1.0), 8, 0)
                par = chain_check(T, np, fi, x, args.trials, capi.FECFRAME_NORMAL, msg, r, c, "demapper oracle (parity unpinned) + ")
            warm(fn)  # (warm again after the seconds the checker took)
            ch.profile(True)
            runs, gruns = [], []; t = timed_median(fn, steps2, shard, dev, runs, gruns)
            ti = ldpc_table_info(fi["table"])
            bl = ldpc_bytes(64800, fi["bch_n"] // 8, ti["links_total"], args.trials)
            b_step = 8 * 21600 + 64800 + bl + fi["bch_n"] // 8 + fi["bch_k"] // 8
            val = world * nf * steps2 / t
            configs["config3"] = {"workload": f"8PSK 3/4 normal: demapper + LDPC (S2_TABLE_B7) + BCH(48600,48408,12), {args.trials} iterations cap, "
                                              f"batch={nf}, noise-only symbols (every frame runs the cap; BCH sees failed frames)",
                                  "value": val, "unit": "frames/s", "coded_gbps": val * 64800 / 1e9, "frames_per_gpu": nf,
                                  "max_trials": args.trials, "steps": steps2, "ms_per_step": t / steps2 * 1e3, "ms_per_step_runs": runs, "gpu_ms_per_step_runs": gruns, "ms_per_step_spread": spread(runs), "parity": par,
                                  "roofline": roofline(ch, bl, nf, None, "config3", args.trials, ti["links_total"]), "step_bytes_per_frame": b_step,
                                  "step_frac_of_hbm_peak": b_step * val / world / 1e9 / HBM_PEAK_GBS}
            ch.close(); del syms
        if "config4" in want:
            ldpc_only("config4", "S2_TABLE_C1", 16384, 25, "QPSK 1/4 short (S2_TABLE_C1, N=16200), 25 iterations cap, batch=16384, noise LLRs")
        if "config5" in want:
            llr_chain("config5", "C9_10", nf, args.trials,
                      f"9/10 normal from LLRs: LDPC (S2_TABLE_B11) + BCH(58320,58192,8), {args.trials} iterations cap, {nf} frames per GPU "
                      f"x {world} GPU(s) (BASELINE: 32768 over 8; the reference has no 32APSK demapper), noise LLRs")
        if "config5_s2x" in want:
            llr_chain("config5_s2x", "C154_180", nf, args.trials,
                      f"S2X 154/180 normal from LLRs: LDPC (S2X_TABLE_B21) + BCH(55440,55248,12), {args.trials} iterations cap, {nf} frames per GPU, noise LLRs")
        out["configs"] = configs

    # ---------------------------------------------------------------- SURVEY 8(d) secondaries of config 2 (one GPU)
    if world == 1 and not args.no_configs and args.input == "noise":
        extras = [c for c in args.only.split(",") if c] or ["config2_awgn", "config3_awgn", "config4_awgn", "config2_host", "config3_host", "demap",
                                                            "device_copy", "host_link", "mapping_ceiling"]
        if "host_link" in extras or "config3_host" in extras:
            trace("host_link")
            out["host_link"] = host_link_bandwidth(capi, local)
        bl50 = ldpc_bytes(N, out_bytes, info["links_total"], args.trials)
        if "config2_awgn" in extras:
            trace("config2_awgn")
            d = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", outputmode=capi.OM_MESSAGE,
                            max_trials=args.trials, group_size=G, max_frames=nf, device=local)
            x = awgn_llr(4242)
            b = torch.empty((nf, out_bytes), dtype=torch.uint8, device=dev)
            r = torch.empty((nf + G - 1) // G, dtype=torch.int32, device=dev)
            par = ldpc_gate(T, np, torch, d, table, x, G, args.trials, stream, gate_full)[0] if gate_on else "skipped"
            fn = lambda: d.work_device(x.data_ptr(), nf, b.data_ptr(), 0, r.data_ptr(), stream)
            warm(fn); d.profile(True)
            runs, gruns = [], []; t = timed_median(fn, steps2, shard, dev, runs, gruns)
            upd = torch.where(r < 0, torch.full_like(r, args.trials), args.trials - r).float()
            mean_upd = float(upd.mean().item())
            bl = ldpc_bytes(N, out_bytes, info["links_total"], mean_upd)
            val = nf * steps2 / t
            rla = roofline(d, bl, nf)
            configs["config2_awgn"] = {
                "workload": f"QPSK 1/2 normal at the operating point: valid codewords, QPSK + AWGN at Es/N0 = {args.esn0} dB, LLR = "
                            f"clamp(rint(2 sqrt(2) y / N0)), cap {args.trials}, batch={nf}, G={G} (at 1.5 dB the genuine reference does "
                            "not converge within 50 updates with this LLR scale: DESIGN.md 7)",
                "value": val, "unit": "frames/s", "coded_gbps": val * N / 1e9, "frames_per_gpu": nf, "max_trials": args.trials,
                "steps": steps2, "ms_per_step": t / steps2 * 1e3, "ms_per_step_runs": runs, "gpu_ms_per_step_runs": gruns, "ms_per_step_spread": spread(runs), "parity": par, "es_n0_db": args.esn0,
                "mean_updates_per_group": mean_upd, "min_updates": float(upd.min().item()), "max_updates": float(upd.max().item()),
                "failed_groups": int((r < 0).sum().item()), "roofline": rla,
                # the whole step (first pass + group resolution + finalize) against the bytes of the updates that ran, and against
                # the never-converging rate scaled by cap / mean updates
                "step_frac_of_hbm_peak": bl * val / 1e9 / HBM_PEAK_GBS,
                "frac_of_proportional_rate": val / (out["value"] * args.trials / max(mean_upd, 1e-9))}
            # The same frames through TWO handles in a software pipeline (enqueue / finish, one stream each, call i finished right before
            # call i + 2 is enqueued): what a double-buffering block does. The tail of a launch -- no workgroup left to dispatch while
            # its last groups finish, ~0.9 ms of a 14 ms call -- then runs under the next call's first groups. Same bits, and the
            # group-synchronous stop has to survive two sweep kernels sharing the GPU (fallback rounds reported).
            trace("config2_awgn pipelined")
            d2 = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", outputmode=capi.OM_MESSAGE,
                             max_trials=args.trials, group_size=G, max_frames=nf, device=local)
            hs = [d, d2]
            bs = [b, torch.empty_like(b)]
            rs = [r, torch.empty_like(r)]
            torch.cuda.synchronize(dev)
            fb0 = d.fallback_rounds + d2.fallback_rounds
            tp, ncalls, pruns = pipelined_two(torch, hs, lambda k, st: hs[k].enqueue_device(x.data_ptr(), nf, bs[k].data_ptr(), 0, rs[k].data_ptr(), st), shard, dev)
            valp = nf * ncalls / tp
            configs["config2_awgn"]["pipelined"] = {
                "what": "two handles, enqueue / finish on one stream each, call i finished right before call i + 2 is enqueued; median of three runs",
                "value": valp, "unit": "frames/s", "calls": ncalls, "ms_per_call_runs": pruns,
                "frac_of_proportional_rate": valp / (out["value"] * args.trials / max(mean_upd, 1e-9)),
                "same_results": bool(torch.equal(rs[0], rs[1]) and torch.equal(bs[0], bs[1])),
                "fallback_rounds": d.fallback_rounds + d2.fallback_rounds - fb0}
            d2.close()
            d.close(); del x
        if "config3_awgn" in extras:
            trace("config3_awgn")
            # SURVEY 8(d) config 3 on the input it names first: valid BCH o LDPC codewords, 8PSK-mapped through the inverse of the block's
            # de-interleaver (lib/xfecframe_demapper_cb_impl.cc:162-176: column c of the 21600 x 3 matrix = LLRs c * 21600 ...), AWGN at
            # Es/N0 = 8.5 dB, N0 supplied per call (:148; the loopback of examples/dvbs2_fec_ber.grc:809-830). Whole-batch gate: demapper
            # restatement (parity unpinned) -> genuine LDPC on all cores -> BCH codec (lib/bch_decoder_bb_impl.cc:94-113).
            fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, "C3_4")
            ti = ldpc_table_info(fi["table"])
            es3 = 8.5
            syms, n0v, msg0 = awgn_8psk_symbols(es3)
            ch = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=args.trials, device=local)
            n0 = torch.tensor([float(n0v)], dtype=torch.float32, device=dev)
            msg = torch.empty((nf, ch.msg_bytes), dtype=torch.uint8, device=dev)
            r = torch.empty((nf + G - 1) // G, dtype=torch.int32, device=dev)
            c = torch.empty(nf, dtype=torch.int32, device=dev)
            fn = lambda: ch.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, msg.data_ptr(), r.data_ptr(), c.data_ptr(), stream)
            fn()
            par = "skipped"
            if gate_on:
                ng = nf if gate_full and T.ref_ldpc() is not None and G == 32 else G
                x = T.oracle_demap(syms[:ng].cpu().numpy().view(np.complex64), n0v, 8, 0)
                par = chain_check(T, np, fi, x, args.trials, capi.FECFRAME_NORMAL, msg, r, c, "demapper oracle (parity unpinned) + ")
            warm(fn)
            sent_ok = bool(np.array_equal(msg.cpu().numpy(), np.tile(msg0, (nf // 64 + 1, 1))[:nf]))
            ch.profile(True)
            runs, gruns = [], []; t = timed_median(fn, steps2, shard, dev, runs, gruns)
            upd = torch.where(r < 0, torch.full_like(r, args.trials), args.trials - r).float()
            mean_upd = float(upd.mean().item())
            bl = ldpc_bytes(ti["N"], fi["bch_n"] // 8, ti["links_total"], mean_upd)
            val = nf * steps2 / t
            cv, cc = torch.unique(c, return_counts=True)
            c3 = configs.get("config3", {}).get("value")
            configs["config3_awgn"] = {
                "workload": f"8PSK 3/4 normal chain from symbols at the operating point: 8PSK-mapped BCH o LDPC codewords + AWGN at Es/N0 = {es3} dB, "
                            f"N0 supplied as input, demapper + LDPC (S2_TABLE_B7, cap {args.trials}) + BCH(48600,48408,12), batch={nf}, G={G}",
                "value": val, "unit": "frames/s", "coded_gbps": val * ti["N"] / 1e9, "frames_per_gpu": nf, "max_trials": args.trials,
                "steps": steps2, "ms_per_step": t / steps2 * 1e3, "ms_per_step_runs": runs, "gpu_ms_per_step_runs": gruns, "ms_per_step_spread": spread(runs), "parity": par, "es_n0_db": es3, "n0": float(n0v),
                "mean_updates_per_group": mean_upd, "min_updates": float(upd.min().item()), "max_updates": float(upd.max().item()),
                "failed_groups": int((r < 0).sum().item()), "decoded_messages_equal_the_sent_ones": sent_ok,
                "bch_corrections_histogram": {str(int(a)): int(b) for a, b in zip(cv.tolist(), cc.tolist())},
                "roofline": roofline(ch, bl, nf),
                "frac_of_proportional_rate": (val / (c3 * args.trials / max(mean_upd, 1e-9))) if c3 else None}
            trace("config3_awgn pipelined")
            ch2 = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=args.trials, device=local)
            hs = [ch, ch2]
            ms2 = [msg, torch.empty_like(msg)]
            cs2 = [c, torch.empty_like(c)]
            tp, ncalls, pruns = pipelined_two(torch, hs, lambda k, st: hs[k].enqueue_device(syms.data_ptr(), nf, n0.data_ptr(), 1, ms2[k].data_ptr(), 0, cs2[k].data_ptr(), st), shard, dev)
            configs["config3_awgn"]["pipelined"] = {
                "what": "two chain handles, enqueue / finish on one stream each, call i finished right before call i + 2 is enqueued; median of three runs",
                "value": nf * ncalls / tp, "unit": "frames/s", "calls": ncalls, "ms_per_call_runs": pruns,
                "frac_of_proportional_rate": (nf * ncalls / tp / (c3 * args.trials / max(mean_upd, 1e-9))) if c3 else None,
                "same_results": bool(torch.equal(ms2[0], ms2[1]) and torch.equal(cs2[0], cs2[1]))}
            ch2.close()
            ch.close(); del syms
        if "config4_awgn" in extras:
            trace("config4_awgn")
            # config 4 at ITS operating point. SURVEY 8(d) names Es/N0 = -1.8 dB; with the demapper's LLR scale (mean |LLR| ~ 1.3, offset
            # beta = 1) the GENUINE reference does not converge there within 25 updates (ret -1 down to -0.5 dB, 19-22 updates at 0.0 dB,
            # 13-14 at 0.5 dB; measured with oracle/_ref in the build container): 0.5 dB, as config 2 runs 0.5 dB above ITS threshold.
            es4, tbl4, fr4, tr4 = 0.5, "S2_TABLE_C1", 16384, 25
            ti = ldpc_table_info(tbl4)
            d = LdpcDecoder(table=tbl4, message_bits=ti["K"], outputmode=capi.OM_MESSAGE, max_trials=tr4, group_size=G, max_frames=fr4, device=local)
            x = qpsk_awgn_llr(tbl4, fr4, es4, 4444)
            b = torch.empty((fr4, d.out_bytes), dtype=torch.uint8, device=dev)
            r = torch.empty((fr4 + G - 1) // G, dtype=torch.int32, device=dev)
            par = ldpc_gate(T, np, torch, d, tbl4, x, G, tr4, stream, gate_full)[0] if gate_on else "skipped"
            fn = lambda: d.work_device(x.data_ptr(), fr4, b.data_ptr(), 0, r.data_ptr(), stream)
            warm(fn); d.profile(True)
            runs, gruns = [], []; t = timed_median(fn, steps2, shard, dev, runs, gruns)
            upd = torch.where(r < 0, torch.full_like(r, tr4), tr4 - r).float()
            mean_upd = float(upd.mean().item())
            val = fr4 * steps2 / t
            c4 = configs.get("config4", {}).get("value")
            configs["config4_awgn"] = {
                "workload": f"QPSK 1/4 short (S2_TABLE_C1) at the operating point: valid codewords, QPSK + AWGN at Es/N0 = {es4} dB, LLR = "
                            f"clamp(rint(2 sqrt(2) y / N0)), cap {tr4}, batch={fr4}, G={G} (the survey's -1.8 dB: the genuine reference does not "
                            "converge within 25 updates below 0.0 dB with this LLR scale)",
                "value": val, "unit": "frames/s", "coded_gbps": val * ti["N"] / 1e9, "frames_per_gpu": fr4, "max_trials": tr4,
                "steps": steps2, "ms_per_step": t / steps2 * 1e3, "ms_per_step_runs": runs, "gpu_ms_per_step_runs": gruns, "ms_per_step_spread": spread(runs), "parity": par, "es_n0_db": es4,
                "mean_updates_per_group": mean_upd, "min_updates": float(upd.min().item()), "max_updates": float(upd.max().item()),
                "failed_groups": int((r < 0).sum().item()),
                "roofline": roofline(d, ldpc_bytes(ti["N"], d.out_bytes, ti["links_total"], mean_upd), fr4),
                "frac_of_proportional_rate": (val / (c4 * tr4 / max(mean_upd, 1e-9))) if c4 else None}
            trace("config4_awgn pipelined")
            d2 = LdpcDecoder(table=tbl4, message_bits=ti["K"], outputmode=capi.OM_MESSAGE, max_trials=tr4, group_size=G, max_frames=fr4, device=local)
            hs = [d, d2]
            bs = [b, torch.empty_like(b)]
            rs = [r, torch.empty_like(r)]
            tp, ncalls, pruns = pipelined_two(torch, hs, lambda k, st: hs[k].enqueue_device(x.data_ptr(), fr4, bs[k].data_ptr(), 0, rs[k].data_ptr(), st), shard, dev)
            configs["config4_awgn"]["pipelined"] = {
                "what": "two handles, enqueue / finish on one stream each, call i finished right before call i + 2 is enqueued; median of three runs",
                "value": fr4 * ncalls / tp, "unit": "frames/s", "calls": ncalls, "ms_per_call_runs": pruns,
                "frac_of_proportional_rate": (fr4 * ncalls / tp / (c4 * tr4 / max(mean_upd, 1e-9))) if c4 else None,
                "same_results": bool(torch.equal(rs[0], rs[1]) and torch.equal(bs[0], bs[1])),
                "fallback_rounds": d.fallback_rounds + d2.fallback_rounds}
            d2.close()
            d.close(); del x
        if "config2_host" in extras:
            trace("config2_host")
            host, fb = host_entry(np, torch, capi, LdpcDecoder, T, dev, local, N, out_bytes, nf, args.trials, G, stream, steps2, (nf, 512))
            pipe_host = host_pipelined(np, torch, capi, LdpcDecoder, dev, local, N, out_bytes, nf, args.trials, G, (nf, 512), host)
            configs["config2_host"] = {"pipelined": pipe_host,
                                       "workload": "dvbs2_ldpc_decode (host buffers in and out: H2D + decode + D2H per synchronous call), "
                                                   f"table B4, cap {args.trials}, noise LLRs; never the headline value", "unit": "frames/s",
                                       "value": host[f"{nf}_pageable"]["frames_per_s"], "calls": host, "fallback_rounds": fb,
                                       "step_frac_of_hbm_peak": bl50 * host[f"{nf}_pageable"]["frames_per_s"] / 1e9 / HBM_PEAK_GBS}
        if "config3_host" in extras:
            trace("config3_host")
            # SURVEY 8(b) fused entry from HOST buffers + 8(d) "end-to-end incl. H2D / D2H": dvbs2_chain_decode on config 3's two inputs
            link = out["host_link"]["hipHostRegister_1_stream"]["h2d"]
            g = torch.Generator(device=dev); g.manual_seed(777 + rank)
            sy = torch.randn((nf, 21600 * 2), generator=g, device=dev) * 0.7071
            worst = chain_host_entry(np, torch, capi, FecChain, dev, local, nf, args.trials, G, stream, sy, np.float32(1.0), (nf, 512), link, steps2,
                                     "noise-only symbols (every frame runs the cap: the kernels bound the call)")
            del sy
            sy, n0v, _ = awgn_8psk_symbols(8.5)
            opp = chain_host_entry(np, torch, capi, FecChain, dev, local, nf, args.trials, G, stream, sy, n0v, (nf, 512), link, steps2,
                                   "operating point: 8PSK-mapped codewords + AWGN at Es/N0 = 8.5 dB (the host link bounds the call)")
            del sy
            configs["config3_host"] = {
                "workload": "dvbs2_chain_decode (HOST symbols in, HOST message bytes out: H2D + demapper + LDPC (S2_TABLE_B7) + BCH + D2H per synchronous call), "
                            f"8PSK 3/4 normal, cap {args.trials}; never the headline value", "unit": "frames/s",
                "value": opp["calls"][f"{nf}_page_locked"]["frames_per_s"], "host_link_h2d_gbs": link, "worst_case": worst, "operating_point": opp}
        if "demap" in extras:
            trace("demap")
            out["demap"] = demap_entry(np, torch, capi, Demapper, T, dev, local, nf, stream, roofline.copy_gbs)
        if "device_copy" in extras:
            out["device_copy"] = device_copy
            out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / out["device_copy"]["read_plus_write_gbs"]
        if "mapping_ceiling" in extras:
            trace("mapping_ceiling")
            # What THIS thread-per-check-row mapping does when nothing orders the rows: S2X_TABLE_B3 (9/20 normal) is B4's hazard-free
            # sibling -- the same check degree 7, the same kernel build, 99 layers, no layer with two entries of one group. Its
            # edge-update rate, measured in this run, projected onto B4's edge count, is the rate B4 would have without its 8 hazard
            # layers: the ceiling of the mapping, beside the HBM roofline of the algorithm.
            sib = "S2X_TABLE_B3"
            ldpc_only("_sib", sib, nf, args.trials, "")
            sc = configs.pop("_sib")
            si = ldpc_table_info(sib)
            eups = sc["value"] * si["links_total"] * args.trials
            proj = eups / (info["links_total"] * args.trials)
            out["roofline"]["mapping_ceiling"] = {
                "hazard_free_sibling": sib, "sibling_kernel": sc["roofline"]["kernel"], "sibling_frames_per_s": sc["value"],
                "sibling_parity": sc["parity"], "edge_updates_per_s": eups, "projected_frames_per_s": proj,
                "projected_frac_of_hbm_peak": b_alg * proj / 1e9 / HBM_PEAK_GBS, "value_frac_of_ceiling": out["value"] / proj,
                "note": "rate of the same kernel build on the hazard-free degree-7 table, per edge update, projected onto B4's edges: "
                        "what is left between `value` and it are B4's 8 hazard layers (ordered lane chains, DESIGN.md 3-4)"}
        out["configs"] = configs

    if world > 1 and not args.no_configs and args.input == "noise":
        # the expected limiter of the sharded job is host feeding, not the GPUs. Every rank runs the host-buffer entry
        # (H2D + decode + D2H per call, call site lib/ldpc_decoder_bb_impl.cc:406-449) at the same time; per-rank and summed rates
        # beside the resident `value`.
        host, fb = host_entry(np, torch, capi, LdpcDecoder, T, dev, local, N, out_bytes, nf, args.trials, G, stream, steps2, (nf,), shard)
        feed = {}
        for mode in ("pageable", "page_locked"):
            per = shard.gather_over_ranks(host[f"{nf}_{mode}"]["frames_per_s"], device=dev)
            res = shard.gather_over_ranks(host[f"{nf}_{mode}"]["resident_frames_per_s"], device=dev)
            feed[mode] = {"per_rank_frames_per_s": per, "sum_frames_per_s": sum(per), "sum_resident_frames_per_s": sum(res),
                          "frac_of_resident": sum(per) / max(sum(res), 1e-9),
                          "link_gbs_consumed_sum": sum(per) * (N + out_bytes + 4.0 / G) / 1e9}
        fbs = shard.gather_over_ranks(float(fb), device=dev)
        out.setdefault("configs", {})["config2_host"] = {
            "workload": f"dvbs2_ldpc_decode from host buffers on all {world} ranks at once ({nf}-frame synchronous calls, table B4, cap "
                        f"{args.trials}, noise LLRs): the host feed of the sharded job; never the headline value",
            "unit": "frames/s", "value": feed["pageable"]["sum_frames_per_s"], "ranks": feed, "fallback_rounds": sum(fbs)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table, N, args.trials)
        print(json.dumps(out))
    shard.finalize()


if __name__ == "__main__":
    main()
