# power draw and shader clock while the benchmark's training steps run (rocm-smi polled beside bench.py)
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-extras > /tmp/bench_power.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
tail -c 400 /tmp/bench_power.json
