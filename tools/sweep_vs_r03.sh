# Development aid: device time of the search over a grid of shapes, this build against round 3's library (tools/exp/libslamhip_r03.so, built
# from the round-3 commit), in one process, bit-exactness checked; flags every shape that got slower by more than 1 %.
L=slam-experiments_amd/lib/libslamhip.so
SH=""
for n in 500 1000 1500 2048 3000 4096 6000 8192 12000 20000 30000 50000 80000 131072 262144; do for m in 16384 24000 40000 65536 100000 200000 400000 1000000; do SH="$SH ${n}x${m}"; done; done
timeout -k 10 1000 python tools/ab_time.py tools/exp/libslamhip_r03.so,$L $SH --rounds 2 --reps 6 --check > gpurun_out/r04/ab_sweep.log 2>&1
python3 - <<'PY'
import re
rows={}
for line in open('gpurun_out/r04/ab_sweep.log'):
    m=re.match(r"(\d+x\d+)\s+(\S+): median\s+([\d.]+) us.*first library's: (\w+)", line)
    if m: rows.setdefault(m.group(1),{})[m.group(2)]=(float(m.group(3)), m.group(4))
bad=[]
for k,v in rows.items():
    if '_r03' in v and 'shipped' in v:
        r=v['shipped'][0]/v['_r03'][0]
        flag = " <-- slower" if r>1.01 else ""
        if v['shipped'][1]!='True': flag+=" TABLE DIFFERS"
        print(f"{k:>16} r03 {v['_r03'][0]:10.1f}  now {v['shipped'][0]:10.1f}  ratio {r:.3f}{flag}")
PY
