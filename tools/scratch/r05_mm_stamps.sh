# round 5: phase stamps of conv133_mm_kernel (diagnostic build libe2e_hip_stamps.so, -DMM_STAMPS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_mm; mkdir -p $O
E2E_LIB_PATH=$PWD/e2enet_medical_amd/csrc/libe2e_hip_stamps.so E2E_MM_STAMPS=1 python tools/kbench.py L0_64x32 L0_32x32d L1_160x64 L2_320x128 2>&1 | grep -v amdgpu | grep "conv133_mm\|fwd\|dgrad" | awk '/conv133_mm/{c[$0]++; if (c[$0]<=1) print; next} {print}' > $O/stamps.txt
python - <<'PY'
import re,collections
seen=collections.OrderedDict()
for l in open('gpurun_out/r05_mm/stamps.txt'):
    m=re.match(r'\[conv133_mm mode (\d) P (\d+) Q (\d+)\]',l)
    if m:
        seen.setdefault(m.groups(),[]).append(l.strip())
    else: print(l.strip())
for k,v in seen.items(): print(v[-1])
PY
