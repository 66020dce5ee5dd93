#!/bin/bash
# Round 5, review item 7: 8-row strips of the 4-pixels-per-lane edge kernel (two halo rows per EIGHT rows instead of per four)
# against the product's 4-row strips, same box: time (interleaved, tools/ab_variants.py) and the counters the review names.
#   variants: python -c "from stereomatching_amd import build; build.build_variants({'e4': [], 'e8': ['-DSM_EDGE4_ROWS=8']})"
#   gpurun -- 'bash tools/ab_edge_strips_pmc.sh'  -> gpurun_out/r05/ab_edge_strips.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05; mkdir -p $OUT
{
for cfg in "C3 1" "C2 1"; do
  set -- $cfg
  AB_EDGES=1 timeout -k 10 200 python3 tools/ab_variants.py $1 $2 9
  AB_STEP=1 timeout -k 10 200 python3 tools/ab_variants.py $1 $2 9 | sed 's/^/step: /'
done
for v in e4 e8; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    d=$OUT/prof_edge_${v}_$(echo $set | cut -d" " -f1); rm -rf $d
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 tools/edge_launch.py C3 --lib stereomatching_amd/variants/$v.so > $d.log 2>&1 || echo "pmc pass failed: $v $set"
  done
done
python3 - <<'PY'
import csv, glob, collections
for v in ("e4", "e8"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r05/prof_edge_{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_edges_ext4" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"{v}: " + ", ".join(f"{k} {sum(x[1:]) / max(1, len(x) - 1):,.0f}" for k, x in sorted(acc.items())))
    if "SQ_WAIT_INST_ANY" in acc and "SQ_WAVE_CYCLES" in acc:
        wa, wc = sum(acc["SQ_WAIT_INST_ANY"][1:]), sum(acc["SQ_WAVE_CYCLES"][1:])
        print(f"{v}: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {wa / wc:.3f}; FETCH x 2 (gfx950 half count) = "
              f"{2 * sum(acc['FETCH_SIZE'][1:]) / max(1, len(acc['FETCH_SIZE']) - 1) / 1024:.2f} MiB per launch of both 4K images (16.6 MB of pixels)")
PY
} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_edge_strips.txt
