# Power / clock evidence for the three hot kernels -> gpurun_out/r05_power/ (summary: profiles/r05_power_clock.txt)
#   1. the same launches on all / 128 / 64 CUs (ROC_GLOBAL_CU_MASK), with the shader clock workgroup 0 saw (e2e_diag_kernel_clock)
#   2. rocm-smi power + clocks sampled while each kernel loops for ~10 s
#   3. the synthetic probes (tools/scratch/clock_probe.hip)
#   4. SQ_BUSY_CYCLES and GRBM_GUI_ACTIVE per kernel / kernel duration (rocprofv3 --pmc + --kernel-trace)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r05_power; rm -rf $O; mkdir -p $O
M64=0xffffffffffffffff
M128=0xffffffffffffffffffffffffffffffff
CASES="L0_64x32 L0_32x32d L1_160x64"
{
echo "== all CUs";  KB_ITERS=50 python tools/kbench.py $CASES 2>&1 | grep "fwd\|dgrad\|wgrad"
echo "== 128 CUs";  ROC_GLOBAL_CU_MASK=$M128 KB_ITERS=50 python tools/kbench.py $CASES 2>&1 | grep "fwd\|dgrad\|wgrad"
echo "== 64 CUs";   ROC_GLOBAL_CU_MASK=$M64  KB_ITERS=50 python tools/kbench.py $CASES 2>&1 | grep "fwd\|dgrad\|wgrad"
} > $O/cu_mask.txt 2>&1
# 2. power / clock samples under a long loop of each kernel
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|mclk|fclk"; sleep 0.3; done ) > $O/smi_samples.txt 2>&1 &
SMI=$!
sleep 2
echo "loop start $(date +%s.%N)" > $O/loop.txt
KB_ITERS=12000 python tools/kbench.py L0_64x32 >> $O/loop.txt 2>&1
echo "loop end $(date +%s.%N)" >> $O/loop.txt
sleep 2
kill $SMI
rocm-smi --showpower --showclocks --showmaxpower > $O/smi_idle.txt 2>&1
# 3. synthetic probes
hipcc --offload-arch=gfx950 -O3 tools/scratch/clock_probe.hip -o /tmp/clock_probe.out 2> $O/probe_build.txt && /tmp/clock_probe.out > $O/probes.txt 2>&1
# 4. counters per kernel
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 $R/tools/kbench.py L0_64x32 > $O/pmc_stdout.txt 2>&1
cd $R
python3 - <<'PY' > gpurun_out/r05_power/pmc_summary.txt 2>&1
import csv, glob, collections
cc = glob.glob('gpurun_out/r05_power/pmc/**/*counter_collection.csv', recursive=True)
kt = glob.glob('gpurun_out/r05_power/pmc/**/*kernel_trace.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d, v in vals.items():
    if d in dur:
        ns, name = dur[d]
        a = agg[name[:60]]
        a[0] += 1; a[1] += ns; a[2] += v.get('SQ_BUSY_CYCLES', 0.0); a[3] += v.get('GRBM_GUI_ACTIVE', 0.0)
print("%-60s %6s %10s %22s %20s" % ("kernel", "n", "avg ms", "SQ_BUSY_CYCLES/32/ns GHz", "GRBM_GUI_ACTIVE/ns GHz"))
for name, (n, ns, sq, gui) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns / n > 50000:
        print("%-60s %6d %10.3f %22.3f %20.3f" % (name, n, ns / n / 1e6, sq / 32 / ns, gui / ns))
PY
tail -30 $O/cu_mask.txt; tail -5 $O/pmc_summary.txt; tail -12 $O/smi_samples.txt; cat $O/probes.txt
