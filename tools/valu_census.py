#!/usr/bin/env python3
"""tools/valu_census.py -> profiles/valu_mix.json

STATIC census of the VALU instructions of every sweep-kernel build (the device code objects inside gr-dvbs2rx_amd/build/ldpc_inst_*.o,
`llvm-objdump -d`), by the issue-rate classes measured on the MI355X with tools/ubench/valu_rate.hip (notes/r04_experiments.md, "VALU rates,
second look"; cycles per wave-instruction per SIMD, eight waves per SIMD, at the clock of that run):
    full     2.65  add / sub / and / or / xor / not / mov / v_lshrrev / v_ashrrev / v_bitop3 / v_add3 / saturating subtract / fp32 add, mul, fma / v_min_u16
    half     4.3   every 32-bit and fp32 min / max / med3 / min3 / max3, v_sad_*, v_bfe, v_perm, v_lshlrev, conversions, SDWA forms, v_pk_*, compares, v_cndmask
    quarter  8.2   v_med3_i16, v_min3_i16, v_add_i16 clamp
bench.py turns the SQ pass's instruction count into ISSUE CYCLES with this mix (roofline.limiter) instead of the flat "4 cycles per
instruction" of round 5. The mix is the whole kernel's text, not the executed path's: layer 0, the syndrome test and rarely taken branches
are in it -- an approximation, said so in the bench line. Run after `make -C gr-dvbs2rx_amd`; ties itself to the digest of csrc/.
"""
import collections
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha256  # noqa: E402

B = "/opt/rocm/lib/llvm/bin/"
RATES = {"full": 2.65, "half": 4.3, "quarter": 8.2}
QUARTER = re.compile(r"^v_(med3_i16|min3_i16|max3_i16|med3_u16|min3_u16|max3_u16)")
HALF = re.compile(r"^v_(min|max|med3|min3|max3|sad|msad|bfe|perm|lshlrev|cvt|pk_|dot|cmp|cmpx|cndmask|alignbit|alignbyte|mul_|mad_|bfi|ffb|bcnt|mbcnt|lshl_add|lshl_or|add_lshl|readlane|writelane|readfirstlane)")
FULL = re.compile(r"^v_(add|sub|subrev|and|or|xor|not|mov|lshrrev|ashrrev|bitop3|add3|or3|xor3|and_or|fma_f32|mul_f32|fmac|nop|accvgpr|swap)")


def classify(op):
    base = op
    if op.endswith("_sdwa"):
        return "half"
    if re.match(r"^v_add_i16|^v_sub_i16", op):
        return "quarter"
    if re.match(r"^v_min_u16|^v_max_u16", base):
        return "full"
    if QUARTER.match(base):
        return "quarter"
    if HALF.match(base):
        return "half"
    if FULL.match(base):
        return "full"
    return "unknown"


def short(mangled):
    m = re.search(r"ldpc_layered_pr_kernelILb([01])ELb([01])E", mangled)  # <W1, V2>
    if m:
        return "ldpc_layered_pr_kernel<w1>" if m.group(1) == "1" else "ldpc_layered_pr_kernel<packed>" if m.group(2) == "1" else "ldpc_layered_pr_kernel"
    m = re.search(r"ldpc_layered_kernelILi(\d+)ELb([01])ELi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])ELb([01])E", mangled)
    if not m:
        return None
    d, timing, minw, v2, solo, chain, hz2, soft = m.groups()
    if timing == "1":
        return None
    if minw != "1":
        return f"ldpc_layered_kernel<{d}, dense>"
    s = f"ldpc_layered_kernel<{d}" + (", packed" if v2 == "1" else "")
    return s + (", solo>" if solo == "1" else ", hz2>" if hz2 == "1" else ", soft>" if soft == "1" else ">")


def main():
    out = {"note": __doc__.split("\n\n")[1] if "\n\n" in __doc__ else "", "csrc_sha256": csrc_sha256(), "rates_cycles_per_wave_instruction_per_simd": RATES,
           "rates_source": "tools/ubench/valu_rate.hip on MI355X, notes/r04_experiments.md (measured at 2.4 GHz)", "kernels": {}}
    unknown_ops = collections.Counter()
    with tempfile.TemporaryDirectory() as td:
        for obj in sorted(glob.glob(os.path.join(ROOT, "gr-dvbs2rx_amd", "build", "ldpc_inst_*.o"))):
            fb, elf = os.path.join(td, "k.fatbin"), os.path.join(td, "k.elf")
            subprocess.check_call([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
            subprocess.check_call([B + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + fb, "--output=" + elf])
            dis = subprocess.run([B + "llvm-objdump", "-d", elf], capture_output=True, text=True, check=True).stdout
            name = None
            counts = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    name = short(m.group(1))
                    counts = out["kernels"].setdefault(name, collections.Counter()) if name else None
                    continue
                if counts is None:
                    continue
                m = re.match(r"^\s+([a-z_0-9]+)", line)
                if not m:
                    continue
                op = m.group(1)
                if op.startswith("v_"):
                    c = classify(op)
                    if c == "unknown":
                        unknown_ops[op] += 1
                    counts[c] += 1
                elif op.startswith("ds_"):
                    counts["lds"] += 1
                elif op.startswith("s_"):
                    counts["salu_smem"] += 1
                elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
                    counts["vmem"] += 1
    for k, c in out["kernels"].items():
        v = c["full"] + c["half"] + c["quarter"] + c["unknown"]
        cyc = (c["full"] + c["unknown"]) * RATES["full"] + c["half"] * RATES["half"] + c["quarter"] * RATES["quarter"]
        out["kernels"][k] = {"valu": v, "full": c["full"], "half": c["half"], "quarter": c["quarter"], "unknown_counted_as_full": c["unknown"],
                             "lds": c["lds"], "salu_smem": c["salu_smem"], "vmem": c["vmem"], "cycles_per_valu_instruction": cyc / max(v, 1)}
    out["unknown_opcodes"] = dict(unknown_ops.most_common(40))
    dst = os.path.join(ROOT, "profiles", "valu_mix.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k in sorted(out["kernels"]):
        e = out["kernels"][k]
        print(f"{k:44s} valu {e['valu']:6d} full {e['full'] / e['valu']:.2f} half {e['half'] / e['valu']:.2f} quarter {e['quarter'] / e['valu']:.3f} "
              f"unknown {e['unknown_counted_as_full'] / e['valu']:.3f} -> {e['cycles_per_valu_instruction']:.2f} cycles / instruction")
    print("unknown:", out["unknown_opcodes"])


if __name__ == "__main__":
    main()
