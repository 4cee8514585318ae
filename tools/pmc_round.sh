#!/bin/bash
# usage (GPU box): bash tools/pmc_round.sh <tag>
# Instruction-mix and LDS counters of the headline step (bench.py --no-graph), two --pmc passes of their own; per-launch
# averages by kernel name with the derived fractions -> gpurun_out/pmc_<tag>_step.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/pmc_${tag}_step
mkdir -p $out/p1 $out/p2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph > $out/p1.out 2> $out/p1.err || { echo pass 1 failed; tail -3 $out/p1.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph > $out/p2.out 2> $out/p2.err || { echo pass 2 failed; tail -3 $out/p2.err; exit 1; }
python3 - <<PY
import csv, collections, glob, json
res = collections.defaultdict(dict)
for d in ("p1", "p2"):
    f = glob.glob("$out/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'sv::' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        for c, x in v.items():
            res[k][c if c != "GRBM_GUI_ACTIVE" or d == "p1" else "GRBM_GUI_ACTIVE_pass2"] = round(sum(x) / len(x))
for k, v in res.items():
    wall = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if wall > 0:
        v["derived"] = {"wall_cycles": round(wall),
                        "valu_issue_frac_at_4cyc": round(v.get("SQ_INSTS_VALU", 0) * 4 / (wall * 1024), 3),
                        "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (wall * 1024), 3)}
json.dump(res, open("$out.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda t: -t[1].get("GRBM_GUI_ACTIVE", 0)):
    print(k, v)
PY
rm -rf $out
