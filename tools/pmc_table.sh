#!/bin/bash
# tools/pmc_table.sh <table:trials:frames> <outdir> -- GPU box: SQ counter passes of the sweep kernel for one table
T=${1:-S2X_TABLE_B21:50:4096}; OUT=${2:-gpurun_out/pmc}
REPO=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $REPO/tools/exp_tables.py $T"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $REPO/$OUT/p1 -o p -- $CMD > $REPO/$OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $REPO/$OUT/p2 -o p -- $CMD > $REPO/$OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT -d $REPO/$OUT/p3 -o p -- $CMD > $REPO/$OUT/p3.log 2>&1
cd $REPO
python - <<PY
import sqlite3, glob, sys
for d in ("p1","p2","p3"):
    for db in glob.glob("$OUT/%s/**/*.db" % d, recursive=True):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if "pmc_event" in t]; info = [t for t in tabs if "info_pmc" in t]; kd=[t for t in tabs if "kernel_dispatch" in t]; ks=[t for t in tabs if "info_kernel_symbol" in t]
        if not pmc: continue
        q = f"select s.kernel_name, i.name, sum(e.value), count(distinct k.dispatch_id) from {pmc[0]} e join {info[0]} i on e.pmc_id=i.id join {kd[0]} k on e.event_id=k.id join {ks[0]} s on k.kernel_id=s.id group by 1,2"
        try:
            for kn, cn, v, n in con.execute(q):
                if "ldpc_layered" in kn: print(f"{kn[:40]:40s} {cn:26s} {v:16.0f} over {n} dispatches")
        except Exception as ex:
            print("query failed", ex, tabs[:8])
PY
