#!/usr/bin/env python3
"""Container only: reads the enumerations of the reference's include/gnuradio/dvbs2rx/dvb_config.h (the integer values the
GNU Radio blocks pass to their constructors) and writes them as DATA to tests/golden/dvb_config_enums.json, so that the
C ABI's DVBS2_* constants, the ctypes binding and the C++ host mirror can be pinned against them on any box."""
import json, os, re, sys
REF = "/root/reference/include/gnuradio/dvbs2rx/dvb_config.h"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = re.sub(r"//[^\n]*", "", open(REF).read())
out = {}
for m in re.finditer(r"enum\s+(\w+)\s*\{([^}]*)\}", text):
    name, body = m.group(1), m.group(2)
    val, d = -1, {}
    for item in [x.strip() for x in body.split(",") if x.strip()]:
        if "=" in item:
            k, v = [y.strip() for y in item.split("=")]
            val = int(v, 0)
        else:
            k, val = item, val + 1
        d[k] = val
    out[name] = d
wanted = ["dvb_standard_t", "dvb_code_rate_t", "dvb_framesize_t", "dvb_constellation_t", "dvb_outputmode_t", "dvb_infomode_t"]
json.dump({"source": "include/gnuradio/dvbs2rx/dvb_config.h (reference v1.4.0)", "enums": {k: out[k] for k in wanted}},
          open(os.path.join(ROOT, "tests", "golden", "dvb_config_enums.json"), "w"), indent=1)
print({k: len(out[k]) for k in wanted})
