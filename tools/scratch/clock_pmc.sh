# shader clock of the probe kernels = SQ_BUSY_CYCLES / 32 SQ instances / kernel duration (rocprofv3: counters + kernel trace)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/clock_pmc; mkdir -p $R/gpurun_out/clock_pmc
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clock_pmc -- $R/tools/scratch/clock_probe.out > $R/gpurun_out/clock_pmc/stdout.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob('gpurun_out/clock_pmc/**/*counter_collection.csv', recursive=True)
kt = glob.glob('gpurun_out/clock_pmc/**/*kernel_trace.csv', recursive=True)
print(cc, kt)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
for d, v in sorted(vals.items(), key=lambda kv: int(kv[0])):
    if d in dur:
        ns, name = dur[d]
        print(d, name[:30], "%.3f ms" % (ns / 1e6), {k: "%.3e" % x for k, x in v.items()}, "SQ_BUSY/32/ns = %.3f GHz" % (v.get('SQ_BUSY_CYCLES', 0) / 32 / ns), "GUI_ACTIVE/ns = %.3f" % (v.get('GRBM_GUI_ACTIVE', 0) / ns))
PY
