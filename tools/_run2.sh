R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_WAIT_IFETCH|SQC_INST" | head -40 > $O/avail_icache.txt
cat $O/avail_icache.txt | cut -c1-160
export ENV=myoHandPoseRandom-v0 B=4096 STEPS=20
rm -rf $O/pq_ic1 $O/pq_ic2
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/pq_ic1 -- python3 $R/tools/prof_step.py > $O/pq_ic1.log 2>&1 || echo "pass1 failed"
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pq_ic2 -- python3 $R/tools/prof_step.py > $O/pq_ic2.log 2>&1 || echo "pass2 failed"
cd $R
python3 - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pq_ic1","gpurun_out/pq_ic2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void step_kernel_w<") or "step_kernel_w<" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items(): print(d.split('/')[-1], k, "n", len(v), "mean per launch %.4g" % (sum(v[5:])/max(1,len(v[5:]))))
PY
