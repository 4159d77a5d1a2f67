#!/bin/bash
# GPU box: memory-side counters per kernel of one bench config (L2 hit rate, L1->L2 read latency, address translation),
# each --pmc set its own run.   usage: [CFG=cfg5] bash tools/run_mem_counters.sh   -> gpurun_out/mem_<cfg>_*/ + digest
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cfg=${CFG:-cfg5}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  out=$root/gpurun_out/mem_${cfg}_$i
  rocprofv3 --pmc $set --output-format csv -d $out -- python3 $root/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2> $out.log
  i=$((i+1))
done
CFG=$cfg python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = os.environ["CFG"]
tot = {}
for f in glob.glob(f"{root}/gpurun_out/mem_{cfg}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ge2e" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 90000:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ge2e::", "")[:28]
        tot.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for name, d in sorted(tot.items()):
    t = {k: max(v) for k, v in d.items()}
    line = f"{cfg} {name:28s}"
    if t.get("TCC_REQ_sum"): line += f" L2 hit {t.get('TCC_HIT_sum', 0) / max(t.get('TCC_HIT_sum', 0) + t.get('TCC_MISS_sum', 0), 1):.3f} (req {t['TCC_REQ_sum']:.3g})"
    if t.get("TCP_TCC_READ_REQ_sum"): line += f" | L1->L2 reads {t['TCP_TCC_READ_REQ_sum']:.3g}, avg latency {t.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / t['TCP_TCC_READ_REQ_sum']:.0f} cyc"
    if t.get("TCP_UTCL1_REQUEST_sum"): line += f" | TLB miss {t.get('TCP_UTCL1_TRANSLATION_MISS_sum', 0) / t['TCP_UTCL1_REQUEST_sum']:.4f}"
    print(line)
    print("    ", {k: f"{v:.4g}" for k, v in sorted(t.items())})
PY
