#!/usr/bin/env python3
"""tools/gen_traffic.py <summary.txt of tools/profile_round.sh> -> profiles/traffic.json

HBM bytes per launch of every BASELINE configuration's dominant kernel from the rocprofv3 PMC passes of ONE bench.py run
(`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gate none --only config3,config4,config5,config5_s2x`; FETCH_SIZE and
WRITE_SIZE in separate passes). Per kernel the counters are sums over all its dispatches in the run, every one a whole batch
(warm-up + timed steps, no parity-gate launches); divided by the pass's own dispatch count = one launch of the configuration's batch. The file records the digest of
gr-dvbs2rx_amd/csrc it was profiled at (bench.csrc_sha256); bench.py reports `traffic` only for exactly that tree. Units KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte read requests at
64 B; calibrated in round 1 against this kernel family's known message byte count). Each entry also carries the SQ pass of the same run
(`sq`: instruction / activity / wait counters summed over the kernel's dispatches + their total duration), from which bench.py computes
`roofline.limiter`."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha256
src = sys.argv[1]
txt = open(src).read()
line = [l for l in txt.split("\n") if l.startswith("{")][-1]
bench = json.loads(line)


def short(rocname):
    """rocprofv3 name -> dvbs2_ldpc_kernel_name() form"""
    if "ldpc_layered_pr_kernel" in rocname:
        m = re.search(r"ldpc_layered_pr_kernel<(\w+)(?:, (\w+))?>", rocname)  # <W1, V2>
        return "ldpc_layered_pr_kernel<w1>" if m and m.group(1) == "true" else "ldpc_layered_pr_kernel<packed>" if m and m.group(2) == "true" else "ldpc_layered_pr_kernel"
    m = re.search(r"ldpc_layered_kernel<(\d+), (\w+), (\d+), (\w+), (\w+), (\w+), (\w+), (\w+)>", rocname)
    if not m:
        return None
    d, timing, minw, v2, solo, chain, hz2, soft = m.groups()
    if minw != "1":
        return f"ldpc_layered_kernel<{d}, dense>"
    s = f"ldpc_layered_kernel<{d}" + (", packed" if v2 == "true" else "")
    s += ", solo>" if solo == "true" else ", hz2>" if hz2 == "true" else ", soft>" if soft == "true" else ">"
    return s


def counters(section):
    out = {}
    m = re.search(r"== \S*" + section + r"/p_results.db\n(.*?)(?:\n== |\Z)", txt, re.S)
    for l in (m.group(1) if m else "").split("\n"):
        mm = re.match(r"(.*?) \{(.*)\}$", l)
        if mm and short(mm.group(1)):
            d = dict(re.findall(r"'(\w+)': (\d+)", mm.group(2)))
            out[short(mm.group(1))] = {k: int(v) for k, v in d.items()}
    return out


fetch, write, sqp = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq")
steps, warm = bench["steps"], bench["warmup"]
G = bench["config"]["group_size"]
runs = {"config2": (bench["roofline"]["kernel"], bench["config"]["frames_per_gpu"], bench["config"]["max_trials"],
                    (warm + steps) * bench["config"]["frames_per_gpu"])}
for name, c in bench.get("configs", {}).items():
    f = c["frames_per_gpu"]
    total = (1 + c["steps"]) * f  # one untimed call + the timed steps (--gate none: no other launch of the kernel)
    runs[name] = (c["roofline"]["kernel"], f, c["max_trials"], total)
entries = []
for name, (kern, frames, trials, total) in runs.items():
    if kern not in fetch or kern not in write:
        print("no counters for", name, kern); continue
    fs, ws = fetch[kern]["FETCH_SIZE"], write[kern]["WRITE_SIZE"]
    # per launch = sum / dispatches of the pass (every dispatch of the run decodes one whole batch: --gate none, no 32-frame gate launches; the
    # number of warm-up calls is time-based since round 5, so the counters' own dispatch counts are what is divided by)
    per_launch = int((2 * fs / fetch[kern]["_dispatches"] + ws / write[kern]["_dispatches"]) * 1024)
    total = frames * fetch[kern]["_dispatches"]
    entries.append({"config": name, "kernel": kern, "frames_per_launch": frames, "max_trials": trials, "fetch_size_kb_raw_sum": fs,
                    "write_size_kb_sum": ws, "frames_in_sum": total, "hbm_bytes_per_launch": per_launch, "source": "profiles/" + os.path.basename(src)})
    if kern in sqp:  # the SQ pass of the same run (sums over its dispatches): bench.py derives `roofline.limiter` from it
        q = sqp[kern]
        entries[-1]["sq"] = {"dispatches": q["_dispatches"], "dur_ns": q["_dur_ns"], **{k: v for k, v in q.items() if k.startswith("SQ_")}}
    print(f"{name:12s} {kern:40s} {per_launch/1e9:8.2f} GB per launch of {frames} frames")
json.dump({"note": __doc__.split("\n\n", 1)[1], "csrc_sha256": csrc_sha256(), "entries": entries}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
