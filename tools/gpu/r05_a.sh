#!/bin/bash
# round 5, first call: which regime do the C2 kernels run in?  (1) power draw and shader clock sampled from sysfs / rocm-smi while the
# bench loops; (2) the same iteration under lowered shader-clock ceilings (if the box lets root set them): a time that does not move
# with the ceiling means the launch is bound by power or memory, not by instruction issue; (3) gather rate against the leading
# dimension / alignment of the gathered rows.
export TMPDIR=/tmp
O=gpurun_out/r05_a; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | grep -v "^$" | head -40 > $O/smi_idle.txt
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $HW" >> $O/smi_idle.txt; ls $HW >> $O/smi_idle.txt 2>&1
sampler() {  # $1 = output file; samples until the file $1.stop exists
  while [ ! -e $1.stop ]; do
    p=$(cat $HW/power1_average 2>/dev/null || cat $HW/power1_input 2>/dev/null); f=$(cat $HW/freq1_input 2>/dev/null)
    echo "$(date +%s.%N) $p $f" >> $1; sleep 0.02
  done
}
run() {  # name, then the command
  name=$1; shift
  rm -f $O/$name.smp $O/$name.smp.stop
  sampler $O/$name.smp & SP=$!
  "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$name', d.get('ms_per_step'), d.get('roofline',{}).get('frac'))
except Exception as e: print('$name', 'no line', l[:200])
" | tee -a $O/lines.txt
  touch $O/$name.smp.stop; wait $SP
  python - <<PY | tee -a $O/lines.txt
import numpy as np
try:
    a=np.loadtxt('$O/$name.smp')
    p=a[:,1]/1e6; f=a[:,2]/1e6
    hi=p>0.5*p.max()
    print('   $name samples %d  power W: max %.0f mean-of-busy %.0f  | sclk MHz: busy mean %.0f min %.0f max %.0f'%(len(p),p.max(),p[hi].mean(),f[hi].mean(),f[hi].min(),f[hi].max()))
except Exception as e: print('   sampler failed', e)
PY
}
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
run c2_default $B --steps 1500 --warmup 20
run c4shard_default $B --workload c4shard --steps 600 --warmup 20
run c1_default $B --workload c1 --steps 1500 --warmup 20
run c3_default $B --workload c3 --steps 300 --warmup 10
for mhz in 2100 1800 1500 1200; do
  rocm-smi --setperfdeterminism $mhz > $O/setclk_$mhz.txt 2>&1; echo "setperfdeterminism $mhz rc=$?" | tee -a $O/lines.txt
  run c2_clk$mhz $B --steps 800 --warmup 20
  run c4shard_clk$mhz $B --workload c4shard --steps 300 --warmup 20
done
rocm-smi --resetperfdeterminism > /dev/null 2>&1
./tools/microbench/gather_align 2>&1 | tee $O/gather_align.txt
