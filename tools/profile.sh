#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the default bench command.
# Writes small summaries to gpurun_out/prof_<tag>/ (raw CSVs stay in /tmp on the box).
TAG=${1:-r02}
ENVN=${2:-ant}
ENVS=${3:-1024}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_$ENVN
RAW=/tmp/prof_raw_$ENVN
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
MMF=${4:-0}   # MM_caching_frequency (0: the environment's examples/cfg/shac value)
SUFFIX=""; [ "$MMF" != "0" ] && SUFFIX="_mm$MMF"
[ "$ENVS" != "1024" ] && [ "$ENVN" = "ant" ] && SUFFIX="_n$ENVS$SUFFIX"
OUT=$REPO/gpurun_out/prof_${TAG}_$ENVN$SUFFIX
RAW=/tmp/prof_raw_$ENVN$SUFFIX
mkdir -p $OUT $RAW
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-extras --env $ENVN --envs-per-gpu $ENVS --mm-freq $MMF"
echo "$CMD" > $OUT/command.txt
python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.csrc_hash())" > $OUT/csrc_hash.txt 2>/dev/null
run() { timeout 240 rocprofv3 --output-format csv "$@" < /dev/null; }
run --kernel-trace --stats -d $RAW/trace -o t -- $CMD > $OUT/trace.log 2>&1
run --pmc FETCH_SIZE --kernel-trace -d $RAW/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
run --pmc WRITE_SIZE --kernel-trace -d $RAW/write -o w -- $CMD > $OUT/write.log 2>&1
run --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $RAW/sq -o s -- $CMD > $OUT/sq.log 2>&1
run --pmc SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/sq3 -o s3 -- $CMD > $OUT/sq3.log 2>&1
run --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace -d $RAW/sq2 -o s2 -- $CMD > $OUT/sq2.log 2>&1
find $RAW -type f -printf "%p %s\n" > $OUT/files.txt
timeout 120 python - > $OUT/summary.txt 2>&1 < /dev/null <<PY
import csv, glob, collections, os
raw="$RAW"
def load(pat):
    rows=[]
    for f in glob.glob(os.path.join(raw,pat), recursive=True):
        rows+=list(csv.DictReader(open(f)))
    return rows
st=load("trace/**/*kernel_stats.csv")
print("== rocprofv3 --kernel-trace --stats : kernel_stats (top 15 by total time) ==")
if st: print(",".join(st[0].keys()))
for r in st[:15]:
    print(",".join(str(v)[:70] for v in r.values()))
for sub in ("fetch","write","sq","sq2","sq3"):
    rows=load(sub+"/**/*counter_collection.csv")
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in rows:
        k=r.get("Kernel_Name","?")[:48]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    print("== pmc pass",sub,": per-launch averages, dsim kernels ==")
    for k in agg:
        if "dsim" in k:
            print(k,{c:round(agg[k][c]/max(cnt[(k,c)],1),1) for c in agg[k]}, "launches", max(cnt[(k,c)] for c in agg[k]))
PY
cat $OUT/summary.txt
tail -2 $OUT/trace.log
